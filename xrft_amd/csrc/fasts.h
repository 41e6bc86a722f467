// fasts.h -- ONE pass over a small real float32 slab: the whole 2-D transform of a 256 x 256 slab on one CU (xrft.power_spectrum over
// the last two axes of (nt, 256, 256) arrays: reference xrft/xrft.py:685-750 -> fft :307-476, detrend.py:100-113; the reference's
// documented workloads are many small slabs).
//
// A 256 x 256 float32 slab is 256 KB = 32768 packed complex values z[i][j'] = x[i][2j'] + i x[i][2j'+1] = 32 per thread of one
// 1024-thread workgroup (csrc/fastr.h does the same for one long row).  The two-pass pipeline of fasty.h moves 16+ bytes per sample
// through memory (the half-spectrum intermediate makes a round trip); here the slab is read once and the power spectrum written once:
//
//   y:  thread (j', i0) holds rows i = i0 + 8 q, q < 32:  DFT32 over q -> k1,  x W_256^(i0 k1),  exchange among the 8 threads of a packed
//       column,  DFT8 over i0 -> k2:  Z[ky = k1 + 32 k2][j'].  A thread's k1 set is closed under k1 -> 32 - k1 ({2c, 32-2c, 2c+1, 31-2c};
//       {0, 16, 1, 31}), so Z[ky] and Z[256 - ky] sit in ONE thread: the split of the packed columns, E = (Z[ky] + conj Z[-ky]) / 2 (column
//       2j'), O = -i (Z[ky] - conj Z[-ky]) / 2 (column 2j'+1), ky = 0..128, needs no exchange.  Rows 0 and 128 are real and travel as ONE
//       complex row E[0] + i E[128]: 128 rows x 256 columns = the same 32768 values.
//   x:  exchange to thread (row, x0) holding x = x0 + 8 q':  DFT32 over q' -> kx1,  x W_256^(x0 kx1),  exchange among the 8 threads of a
//       row (adjacent lanes),  DFT8 -> kx2:  F[row][kx = kx1 + 32 kx2]; the packed row 0 is split the same way (kx sets closed, too).
//   out: |F|^2 scale staged as 129 float rows in LDS, then every output row leaves whole: row ky rotated by the fftshift, row -ky reversed.
//   The plane of detrend='linear' (xrft/detrend.py:100-113) is exact and local: the whole slab is in the workgroup's registers (float64 sums,
//   wave shuffles, one LDS table added in wave order: bit-reproducible).
// Every exchange moves the slab's 256 KB through the LDS in two halves (<= 139 KB with the padding that makes the 8- and 16-byte accesses
// conflict-free); whole waves write, whole waves read.
#pragma once
#include "fasty.h"  // xrft_store_nt
#include "fastr.h"

namespace xrft {

struct FastS {
    const float* in;    // [slabs][256][256] float32
    float* out;         // [slabs][256][256] float32 power spectrum
    const cf* tw;       // W_256^k, k < 256
    const float* win_y; // 256 samples or null
    const float* win_x;
    long long nslabs;
    int detrend;        // 0 none, 1 constant, 2 linear (plane)
    int shift_y, shift_x;  // 0 or 128
    float scale;
};

constexpr int kFastSThreads = 1024;
constexpr int kFastSLdsElems = 17408;                                   // complex64 elements: 128 * 136 (exchange 2) -- the largest
constexpr size_t kFastSLds = (size_t)kFastSLdsElems * 8 + 16 * 3 * 8;  // + the detrend sums of 16 waves

// member w (0..3) of the k1 set of class c8 (0..7): closed under k -> (32 - k) mod 32; members (0, 1) and (2, 3) are partners
__device__ __forceinline__ int fasts_k1(int c8, int w) {
    if (c8 == 0) return w == 0 ? 0 : w == 1 ? 16 : w == 2 ? 1 : 31;
    return w == 0 ? 2 * c8 : w == 1 ? 32 - 2 * c8 : w == 2 ? 2 * c8 + 1 : 31 - 2 * c8;
}
// the inverse, for a compile-time k1: class and member
constexpr int fasts_c8(int k1) { return k1 == 0 || k1 == 16 || k1 == 1 || k1 == 31 ? 0 : (k1 % 2 == 0 ? (k1 < 16 ? k1 / 2 : (32 - k1) / 2) : (k1 < 16 ? (k1 - 1) / 2 : (31 - k1) / 2)); }
constexpr int fasts_w(int k1) { return k1 == 0 ? 0 : k1 == 16 ? 1 : k1 == 1 ? 2 : k1 == 31 ? 3 : (k1 % 2 == 0 ? (k1 < 16 ? 0 : 1) : (k1 < 16 ? 2 : 3)); }

// the second radix stage of an axis: registers b[w * 8 + i0] (i0 < 8) -> DFT8 over i0 -> b[w * 8 + k2]
__device__ __forceinline__ void fasts_dft8x4(cf* b) {
#pragma unroll
    for (int w = 0; w < 4; ++w) dft8<float>(b + 8 * w);
}

// Split of a packed pair of real sequences held as Z[k], k = k1 + 32 k2, in registers b[w * 8 + k2] of ONE thread (class c8).  On return,
// for every member w and k2 < 4:   b[w * 8 + k2] = E[idx] (the even sequence) and b[(w ^ 1) * 8 + 7 - k2] = O[idx] (the odd one) at the lower
// index idx = k1(c8, w) + 32 k2 < 128 -- the same register positions in every class; idx = 0 (class 0, member 0, k2 = 0) carries the two REAL
// samples 0 and 128 of each sequence packed: E[0] + i E[128] and O[0] + i O[128].
//   E = (Zl + conj Zu) / 2,  O = -i (Zl - conj Zu) / 2   (Zl = Z[idx], Zu = Z[256 - idx])
__device__ __forceinline__ void fasts_pair(cf zl, cf zu, cf& e, cf& o) {
    e = mk<float>(0.5f * (zl.re + zu.re), 0.5f * (zl.im - zu.im));
    o = mk<float>(0.5f * (zl.im + zu.im), 0.5f * (zu.re - zl.re));
}
__device__ __forceinline__ void fasts_split(cf* b, int c8) {
    // members 2, 3 (k1 = 2 c8 + 1 and 31 - 2 c8, partners): k = k1 + 32 k2 pairs with (32 - k1) + 32 (7 - k2) = member w ^ 1, register 7 - k2
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        fasts_pair(b[16 + k2], b[24 + 7 - k2], b[16 + k2], b[24 + 7 - k2]);
        fasts_pair(b[24 + k2], b[16 + 7 - k2], b[24 + k2], b[16 + 7 - k2]);
    }
    if (c8 != 0) {  // members 0, 1 (k1 = 2 c8 and 32 - 2 c8): the same rule
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            fasts_pair(b[k2], b[8 + 7 - k2], b[k2], b[8 + 7 - k2]);
            fasts_pair(b[8 + k2], b[7 - k2], b[8 + k2], b[7 - k2]);
        }
    } else {
        // member 0 is k1 = 0: k = 32 k2 pairs with 32 (8 - k2), k = 0 and 128 are real; member 1 is k1 = 16: k = 16 + 32 k2 pairs with
        // 16 + 32 (7 - k2), the SAME member.  The results move to the positions of the general rule.
        cf e0[4], o0[4], e1[4], o1[4];
        e0[0] = mk<float>(b[0].re, b[4].re);
        o0[0] = mk<float>(b[0].im, b[4].im);
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2) fasts_pair(b[k2], b[8 - k2], e0[k2], o0[k2]);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) fasts_pair(b[8 + k2], b[15 - k2], e1[k2], o1[k2]);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { b[k2] = e0[k2]; b[15 - k2] = o0[k2]; b[8 + k2] = e1[k2]; b[7 - k2] = o1[k2]; }
    }
}

__global__ void __launch_bounds__(kFastSThreads) fasts_power_kernel(FastS p) {
    constexpr int N = 256, T = kFastSThreads, NXP = N / 2;
    XRFT_DYN_SMEM(smem_raw);
    cf* L = reinterpret_cast<cf*>(smem_raw);
    float* Lf = reinterpret_cast<float*>(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw + (size_t)kFastSLdsElems * 8);  // [16 waves][3]
    for (long long slab = blockIdx.x; slab < p.nslabs; slab += gridDim.x) {
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);  // (nothing derived from the thread index is hoisted out of the slab loop and spilled: fastr.h)
        const int jp = tid & (NXP - 1), i0 = tid >> 7;  // packed column, first row
        const cf* __restrict__ src = reinterpret_cast<const cf*>(p.in + (size_t)slab * N * N) + tid;
        cf a[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) a[q] = src[q * T];  // rows i0 + 8 q, columns 2 jp, 2 jp + 1
        if (p.detrend) {
            // S0 = sum x, Si = sum (i - ibar) x, Sj = sum (j - jbar) x over the slab, float64.  With u_q = x[i][2jp] + x[i][2jp+1]:
            //   Si = (i0 - ibar) sum u_q + 8 sum q u_q,   Sj = (2 jp - jbar) sum u_q + sum x[i][2jp+1]
            constexpr double BAR = 0.5 * (N - 1);
            double U = 0.0, V = 0.0, I = 0.0;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const double u = (double)a[q].re + (double)a[q].im;
                U += u;
                V = fma((double)q, u, V);
                I += (double)a[q].im;
            }
            double s0 = U, si = fma((double)i0 - BAR, U, 8.0 * V), sj = fma((double)(2 * jp) - BAR, U, I);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); si += __shfl_xor(si, m); sj += __shfl_xor(sj, m); }
            if ((tid & 63) == 0) { red[(tid >> 6) * 3] = s0; red[(tid >> 6) * 3 + 1] = si; red[(tid >> 6) * 3 + 2] = sj; }
            __syncthreads();
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;  // (`red` is next written a slab later, behind the barriers of the exchanges)
#pragma unroll
            for (int w = 0; w < T / 64; ++w) {
                t0 += red[3 * w]; t1 += red[3 * w + 1]; t2 += red[3 * w + 2];
                if ((w & 3) == 3) fastr_sched_fence();  // (four waves' sums at a time: all 48 values at once are 96 registers beside the slab's 64)
            }
            constexpr double INV_N2 = 1.0 / ((double)N * N), INV_S = 12.0 / ((double)N * N * ((double)N * N - 1.0));  // 1 / sum (i - ibar)^2 over the slab
            const double c1 = p.detrend == 2 ? t1 * INV_S : 0.0, c2 = p.detrend == 2 ? t2 * INV_S : 0.0;
            const double l0 = t0 * INV_N2 + c1 * ((double)i0 - BAR) + c2 * ((double)(2 * jp) - BAR), dl = 8.0 * c1;  // the plane at (i0 + 8 q, 2 jp): l0 + dl q
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                XRFT_OPAQUE(a[q].re); XRFT_OPAQUE(a[q].im);
                const double lq = fma(dl, (double)q, l0);
                a[q] = mk<float>((float)((double)a[q].re - lq), (float)((double)a[q].im - (lq + c2)));
            }
        }
        if (p.win_y) {  // wy[i] wx[j]: the row factors in two batches of 16 beside the slab's registers
            const cf wx = reinterpret_cast<const cf*>(p.win_x)[jp];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float wy[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) wy[q] = p.win_y[i0 + 8 * (16 * g + q)];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[16 * g + q] = mk<float>(a[16 * g + q].re * (wy[q] * wx.re), a[16 * g + q].im * (wy[q] * wx.im));
                fastr_sched_fence();
            }
        }
        // ---- y, stage 1: over q -> k1, x W_256^(i0 k1)   (the fences keep a phase's address arithmetic and table loads out of the one before it:
        // beside the slab's 64 registers there is room for one phase's temporaries, not two)
        fastr_sched_fence();
        dft32f(a);
        fastr_sched_fence();
        twiddle32f(a, p.tw[i0]);
        fastr_sched_fence();
        // ---- exchange 1: writer (jp, i0) registers k1 -> reader (jp, c) registers (w, i0'): element (jp, c, w, i0) of half h (i0 in [4h, 4h + 4))
        // at jp * 129 + c * 16 + w * 4 + (i0 - 4h).  Lane strides of 129 elements = 2 dwords (mod 32 and mod 64): conflict-free.
        const int c = tid >> 7;  // the reader's class: the same thread set {jp + 128 m} serves a packed column before and after
        cf b[32];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if ((i0 >> 2) == h) {
                cf* dst = L + jp * 129 + (i0 & 3);
#pragma unroll
                for (int k1 = 0; k1 < 32; ++k1) dst[fasts_c8(k1) * 16 + fasts_w(k1) * 4] = a[k1];
            }
            __syncthreads();
            const cf* s = L + jp * 129 + c * 16;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[w * 8 + 4 * h + e] = s[w * 4 + e];
        }
        // ---- y, stage 2: DFT8 over i0 -> k2; then the packed columns are split in place
        fastr_sched_fence();
        fasts_dft8x4(b);
        fastr_sched_fence();
        fasts_split(b, c);
        fastr_sched_fence();
        // ---- exchange 2: (jp, c) holds rows {k1 + 32 k2, k2 < 4} of columns 2 jp (lower registers) and 2 jp + 1 (their partners) -> reader
        // (row, x0) holding x = x0 + 8 q'.  Half h carries the columns x in [128 h, 128 h + 128): element (row, x) at row * 136 + (x - 128 h);
        // writers = the threads with jp in [64 h, 64 h + 64) (whole waves), 16 x 16 bytes each; readers all, q' in [16 h, 16 h + 16).
        const int row = tid >> 3, x0 = tid & 7;  // after the exchange: tid = x0 + 8 row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if ((jp >> 6) == h) {
                cf* dst = L + 2 * (jp & 63);
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) {
                        // the even sequence's value at row idx (column 2 jp) and the odd one's (column 2 jp + 1): 16 bytes
                        const int r = fasts_k1(c, w) + 32 * k2;
                        // (two 8-byte stores: a 16-byte one wants its four registers adjacent -- copies beside a full register file)
                        dst[r * 136] = b[w * 8 + k2];
                        dst[r * 136 + 1] = b[(w ^ 1) * 8 + 7 - k2];
                    }
            }
            __syncthreads();
            const cf* s = L + row * 136 + x0;
#pragma unroll
            for (int q = 0; q < 16; ++q) a[16 * h + q] = s[8 * q];
        }
        // ---- x, stage 1: over q' -> kx1, x W_256^(x0 kx1)
        fastr_sched_fence();
        dft32f(a);
        fastr_sched_fence();
        twiddle32f(a, p.tw[x0]);
        fastr_sched_fence();
        // ---- exchange 3: within the 8 adjacent lanes of a row: writer (row, x0) registers kx1 -> reader (row, c') registers (w, x0).  Element
        // (row, c', w, x0) at (row - 64 h) * 264 + c' * 33 + w * 8 + x0; the rows of half h are the threads [512 h, 512 h + 512).
        const int cx = tid & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if ((tid >> 9) == h) {
                cf* dst = L + (row & 63) * 264 + x0;
#pragma unroll
                for (int k1 = 0; k1 < 32; ++k1) dst[fasts_c8(k1) * 33 + fasts_w(k1) * 8] = a[k1];
            }
            __syncthreads();
            if ((tid >> 9) == h) {  // (into `a` itself: a thread writes and reads in ONE half, its old values are dead when the new ones arrive)
                const cf* s = L + (row & 63) * 264 + cx * 33;
#pragma unroll
                for (int e = 0; e < 32; ++e) a[e] = s[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) b[e] = a[e];
        // ---- x, stage 2: DFT8 over x0 -> kx2: b[w * 8 + kx2] = F[row][kx1(cx, w) + 32 kx2]; the packed row 0 is split (rows 0 and 128)
        fastr_sched_fence();
        fasts_dft8x4(b);
        fastr_sched_fence();
        if (row == 0) fasts_split(b, cx);
        fastr_sched_fence();
        // ---- |F|^2 scale, staged as float rows: row r (0 <= r <= 128) at r * 257 + kx
        __syncthreads();
        if (row != 0) {
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const cf v = b[w * 8 + k2];
                    Lf[row * 257 + fasts_k1(cx, w) + 32 * k2] = (v.re * v.re + v.im * v.im) * p.scale;
                }
        } else {
            // the packed row was split: for member w, k2 < 4 the lower register holds F[0][idx], its partner F[128][idx], idx = kx1 + 32 k2
            // (|F[.][256 - idx]| = |F[.][idx]|: both samples are written); idx = 0 carries the four real corner samples
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int idx = fasts_k1(cx, w) + 32 * k2;
                    const cf e = b[w * 8 + k2], o = b[(w ^ 1) * 8 + 7 - k2];
                    if (idx == 0) {  // e = F[0][0] + i F[0][128], o = F[128][0] + i F[128][128]
                        Lf[0] = e.re * e.re * p.scale; Lf[128] = e.im * e.im * p.scale;
                        Lf[128 * 257] = o.re * o.re * p.scale; Lf[128 * 257 + 128] = o.im * o.im * p.scale;
                    } else {
                        const float pe = (e.re * e.re + e.im * e.im) * p.scale, po = (o.re * o.re + o.im * o.im) * p.scale;
                        Lf[idx] = pe; Lf[256 - idx] = pe;
                        Lf[128 * 257 + idx] = po; Lf[128 * 257 + 256 - idx] = po;
                    }
                }
        }
        __syncthreads();
        // ---- every output row whole: row ky (<= 128) rotated by the fftshift, row 256 - ky reversed (F[-ky][-kx] = conj F[ky][kx])
        float* __restrict__ o = p.out + (size_t)slab * N * N;
        for (int e = tid; e < N * N / 4; e += T) {
            const int orow = e >> 6, ch = e & 63;              // output row, float4 chunk
            const int ky = (orow - p.shift_y) & (N - 1);       // unshifted frequency index of this output row
            const bool mir = ky > 128;
            const float* r = Lf + (mir ? N - ky : ky) * 257;
            const int kx0 = (4 * ch - p.shift_x) & (N - 1);   // unshifted kx of the chunk's first sample
            F4 v;
            if (!mir) { v.x = r[kx0]; v.y = r[kx0 + 1]; v.z = r[kx0 + 2]; v.w = r[kx0 + 3]; }
            else { v.x = r[(N - kx0) & (N - 1)]; v.y = r[N - kx0 - 1]; v.z = r[N - kx0 - 2]; v.w = r[N - kx0 - 3]; }
            xrft_store_nt(o + (size_t)orow * N + 4 * ch, v);
        }
    }
}

}  // namespace xrft

// fasts.h -- ONE pass over a small real float32 slab: the whole 2-D transform of a (64 | 128 | 256) x (64 | 128 | 256) slab inside one
// workgroup (xrft.power_spectrum over the last two axes of (nt, ny, nx) arrays: reference xrft/xrft.py:685-750 -> fft :307-476,
// detrend.py:100-113; the reference's documented workloads are many small slabs).
//
// A 256 x 256 float32 slab is 256 KB = 32768 packed complex values z[i][j'] = x[i][2j'] + i x[i][2j'+1] = 32 per thread of one 1024-thread
// workgroup (csrc/fastr.h does the same for one long row); a 128 x 128 slab is 32 per thread of 256 threads, four workgroups per CU.  The
// two-pass pipeline of fasty.h moves 16+ bytes per sample through memory (the half-spectrum intermediate makes a round trip) and starts
// at 256 points per axis; here the slab is read once and the power spectrum written once.  NY = 32 RY, NX = 32 RX, RY, RX in {2, 4, 8}:
//
//   y:  thread (j', i0) holds rows i = i0 + RY q, q < 32:  DFT32 over q -> k1,  x W_NY^(i0 k1),  exchange among the RY threads of a packed
//       column,  DFT_RY over i0 -> k2:  Z[ky = k1 + 32 k2][j'].  A thread's k1 values come in sets of four closed under k1 -> 32 - k1
//       ({2c, 32-2c, 2c+1, 31-2c}; {0, 16, 1, 31}), so Z[ky] and Z[NY - ky] sit in ONE thread: the split of the packed columns,
//       E = (Z[ky] + conj Z[-ky]) / 2 (column 2j'), O = -i (Z[ky] - conj Z[-ky]) / 2 (column 2j'+1), ky = 0..NY/2, needs no exchange.  Rows 0
//       and NY/2 are real and travel as ONE complex row E[0] + i E[NY/2]: NY/2 rows x NX columns = the same number of values.
//   x:  exchange to thread (row, x0) holding x = x0 + RX q':  DFT32 over q' -> kx1,  x W_NX^(x0 kx1),  exchange among the RX threads of a
//       row (adjacent lanes),  DFT_RX -> kx2:  F[row][kx = kx1 + 32 kx2]; the packed row 0 is split the same way (the kx sets are closed, too).
//   out: |F|^2 scale staged as NY/2 + 1 float rows in LDS, then every output row leaves whole: row ky rotated by the fftshift, row -ky reversed.
//   The plane of detrend='linear' (xrft/detrend.py:100-113) is exact and local: the whole slab is in the workgroup's registers (float64 sums,
//   wave shuffles, one LDS table added in wave order: bit-reproducible).
// Every exchange moves the slab through the LDS in two halves (136 KB at 256 x 256, 34 KB at 128 x 128, with the padding that makes the
// 8-byte accesses conflict-free); whole waves write, whole waves read.
#pragma once
#include "fasty.h"  // xrft_store_nt
#include "fastr.h"

namespace xrft {

struct FastS {
    const float* in;    // [slabs][NY][NX] float32
    float* out;         // [slabs][NY][NX] float32 power spectrum
    const cf* tw_y;     // W_NY^k, k < NY
    const cf* tw_x;     // W_NX^k, k < NX
    const float* win_y; // NY samples or null (then win_x is null, too)
    const float* win_x;
    long long nslabs;
    int detrend;        // 0 none, 1 constant, 2 linear (plane)
    int shift_y, shift_x;  // 0 or N/2
    float scale;
    // radial sums (xrft.isotropize, xrft/xrft.py:895-906, 993-1004) of a RADIAL bin map, taken from the staged rows: no partial tables, no
    // atomics, one float64 sum per bin in a fixed order
    double* iso;                   // [slabs][nbins]
    const unsigned short* tfirst;  // [NY/2 + 1][nbins + 1]: the smallest |kx| <= NX/2 of row ky whose bin is >= b (NX/2 + 1 if none)
    int nbins;                     // <= threads of the workgroup
    // complex output (xrft.fft / dft): F scale x the true-phase factors, indexed by unshifted frequency (xrft.py:462-469; an ifftshifted input
    // is the sign (-1)^k folded into the tables)
    const cf* ph_y;
    const cf* ph_x;
    int ph_on;
    int stagger;  // a resident set walking the slabs: start delay of workgroup class c = (block / 8) % classes (fastr.h fastr_stagger; XRFTHIP_FASTS_STAGGER)
};

constexpr size_t fasts_max(size_t a, size_t b) { return a > b ? a : b; }
template <int RY, int RX> struct SGeom {
    static_assert((RY == 2 || RY == 4 || RY == 8) && (RX == 2 || RX == 4 || RX == 8), "64, 128 or 256 points per axis");
    static constexpr int NY = 32 * RY, NX = 32 * RX, NXP = NX / 2, NROW = NY / 2;
    static constexpr int T = RY * NXP;            // threads = packed values / 32
    static constexpr int NW = T / 64 < 1 ? 1 : T / 64;
    static constexpr int KGY = 8 / RY, KGX = 8 / RX;  // sets of four k1 values per thread
    static constexpr int P1 = 16 * RY + 1;        // exchange 1: elements per packed column and half (odd: conflict-free lane stride)
    static constexpr int PX = 17 * RX;            // exchange 2: elements per row and half (NX/2 + RX)
    static constexpr int P3 = 33 * RX;            // exchange 3: elements per row
    static constexpr int PF = NX + 1;             // staged float rows
    static constexpr size_t E1 = (size_t)NXP * P1 * 8, E2 = (size_t)NROW * PX * 8, E3 = (size_t)(NROW / 2) * P3 * 8, EF = (size_t)(NROW + 1) * PF * 4;
    static constexpr size_t EC = (size_t)(NROW / 2 + 1) * PF * 8;  // complex output: half of the rows (+ the Nyquist row) staged at a time
    static constexpr size_t LDS_MAIN = (fasts_max(fasts_max(fasts_max(E1, E2), fasts_max(E3, EF)), EC) + 15) & ~(size_t)15;
    static constexpr size_t LDS = LDS_MAIN + (size_t)NW * 3 * 8;  // + the detrend sums per wave
    static constexpr size_t EFA = (EF + 15) & ~(size_t)15;        // radial sums: the per-chunk partial sums (one float64 per thread) behind the staged rows
    static constexpr size_t LDS_ISO = fasts_max(LDS_MAIN, EFA + (size_t)T * 8) + (size_t)NW * 3 * 8;
    // waves per SIMD asked of the compiler: 4 (128 registers) where two workgroups per CU need it (512 threads) and at 1024 threads; the small
    // workgroups take the 152 registers the kernel wants without spilling and run three waves per SIMD
    static constexpr int WPS = T >= 512 ? 4 : 3;
    static constexpr int PER_CU = T >= 512 ? 2048 / T / 2 : 12 / (T / 64);
};

// member w (0..3) of the k1 set of class c8 (0..7): closed under k -> (32 - k) mod 32; members (0, 1) and (2, 3) are partners
__device__ __forceinline__ int fasts_k1(int c8, int w) {
    if (c8 == 0) return w == 0 ? 0 : w == 1 ? 16 : w == 2 ? 1 : 31;
    return w == 0 ? 2 * c8 : w == 1 ? 32 - 2 * c8 : w == 2 ? 2 * c8 + 1 : 31 - 2 * c8;
}
// the inverse, for a compile-time k1: class and member
constexpr int fasts_c8(int k1) { return k1 == 0 || k1 == 16 || k1 == 1 || k1 == 31 ? 0 : (k1 % 2 == 0 ? (k1 < 16 ? k1 / 2 : (32 - k1) / 2) : (k1 < 16 ? (k1 - 1) / 2 : (31 - k1) / 2)); }
constexpr int fasts_w(int k1) { return k1 == 0 ? 0 : k1 == 16 ? 1 : k1 == 1 ? 2 : k1 == 31 ? 3 : (k1 % 2 == 0 ? (k1 < 16 ? 0 : 1) : (k1 < 16 ? 2 : 3)); }

// Registers of a thread after an axis' exchange: b[(g * 4 + w) * R + j], g < 8 / R (its sets of four), w < 4 (member), j < R.
// the second radix stage of an axis: DFT_R over j -> k2
template <int R> __device__ __forceinline__ void fasts_dft_r(cf* b) {
#pragma unroll
    for (int gw = 0; gw < 32 / R; ++gw) dft_r<float, R>(b + R * gw);
}

//   E = (Zl + conj Zu) / 2,  O = -i (Zl - conj Zu) / 2   (Zl = Z[idx], Zu = Z[N - idx])
__device__ __forceinline__ void fasts_pair(cf zl, cf zu, cf& e, cf& o) {
    e = mk<float>(0.5f * (zl.re + zu.re), 0.5f * (zl.im - zu.im));
    o = mk<float>(0.5f * (zl.im + zu.im), 0.5f * (zu.re - zl.re));
}
// Split of a packed pair of real sequences held as Z[k], k = k1 + 32 k2 (N = 32 R points), in the registers of ONE thread (class c: its
// sets are c8 = c (8 / R) + g).  On return, for every set g, member w and k2 < R/2:   b[(4g + w) R + k2] = E[idx] (the even sequence) and
// b[(4g + (w ^ 1)) R + R - 1 - k2] = O[idx] (the odd one) at the lower index idx = k1(c8, w) + 32 k2 < N/2 -- the same register positions in
// every class; idx = 0 (class 0, set 0, member 0, k2 = 0) carries the two REAL samples 0 and N/2 of each sequence packed: E[0] + i E[N/2]
// and O[0] + i O[N/2].
template <int R> __device__ __forceinline__ void fasts_split(cf* b, int c) {
    constexpr int KG = 8 / R, H = R / 2;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        cf* s = b + 4 * g * R;
        // members 2, 3 (k1 = 2 c8 + 1 and 31 - 2 c8, partners): k = k1 + 32 k2 pairs with (32 - k1) + 32 (R - 1 - k2) = member w ^ 1, register R - 1 - k2
#pragma unroll
        for (int k2 = 0; k2 < H; ++k2) {
            fasts_pair(s[2 * R + k2], s[3 * R + R - 1 - k2], s[2 * R + k2], s[3 * R + R - 1 - k2]);
            fasts_pair(s[3 * R + k2], s[2 * R + R - 1 - k2], s[3 * R + k2], s[2 * R + R - 1 - k2]);
        }
        if (g != 0 || c != 0) {  // members 0, 1 (k1 = 2 c8 and 32 - 2 c8): the same rule
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) {
                fasts_pair(s[k2], s[R + R - 1 - k2], s[k2], s[R + R - 1 - k2]);
                fasts_pair(s[R + k2], s[R - 1 - k2], s[R + k2], s[R - 1 - k2]);
            }
        } else {
            // set 0 of class 0: member 0 is k1 = 0: k = 32 k2 pairs with 32 (R - k2), k = 0 and N/2 are real; member 1 is k1 = 16:
            // k = 16 + 32 k2 pairs with 16 + 32 (R - 1 - k2), the SAME member.  The results move to the positions of the general rule.
            cf e0[H], o0[H], e1[H], o1[H];
            e0[0] = mk<float>(s[0].re, s[H].re);
            o0[0] = mk<float>(s[0].im, s[H].im);
#pragma unroll
            for (int k2 = 1; k2 < H; ++k2) fasts_pair(s[k2], s[R - k2], e0[k2], o0[k2]);
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) fasts_pair(s[R + k2], s[2 * R - 1 - k2], e1[k2], o1[k2]);
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) { s[k2] = e0[k2]; s[2 * R - 1 - k2] = o0[k2]; s[R + k2] = e1[k2]; s[R - 1 - k2] = o1[k2]; }
        }
    }
}

// ISO: 0 the power spectrum; 1 the spectrum and its radial sums; 2 the radial sums only (XRFTHIP_NO_SPECTRUM_OUT).  MODE 1: power spectrum;
// 0: the complex spectrum (xrft.fft / dft; ISO = 0)
template <int RY, int RX, int ISO = 0, int MODE = 1>
__global__ void __launch_bounds__((SGeom<RY, RX>::T), (SGeom<RY, RX>::WPS)) fasts_power_kernel(FastS p) {
    static_assert(MODE == 1 || ISO == 0, "radial sums are of power spectra");
    typedef SGeom<RY, RX> G;
    constexpr int NY = G::NY, NX = G::NX, T = G::T, NXP = G::NXP, NROW = G::NROW, KGY = G::KGY, KGX = G::KGX;
    constexpr int P1 = G::P1, PX = G::PX, P3 = G::P3, PF = G::PF, HY = RY / 2, HX = RX / 2;
    XRFT_DYN_SMEM(smem_raw);
    cf* L = reinterpret_cast<cf*>(smem_raw);
    float* Lf = reinterpret_cast<float*>(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw + (ISO ? G::LDS_ISO : G::LDS) - (size_t)G::NW * 3 * 8);  // [waves][3]
    // The power-spectrum forms (PRE) leave a slab's staged rows in the LDS and emit them at the TOP of the next trip, behind the loads of the next slab:
    // a resident set walking the slabs (gridDim.x < nslabs; the default for long batches of 256 x 256 slabs, ONE workgroup per CU) has the 256 KB of loads
    // in flight beside the 256 KB of stores, and no register value crosses the loop's back edge.  (The complex form stages its result in halves and
    // emits it in place.)
    constexpr bool PRE = MODE == 1;
    // ---- the staged rows of slab `sl` leave: radial sums, then every output row whole
    auto emit = [&](long long sl) {
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);
        if (ISO) {
            // Radial sums straight from the staged rows.  In row ky the bin b of a radial map covers |kx| in [first[ky][b], first[ky][b + 1]):
            // the samples kx = |kx| and kx = NX - |kx|; a row 0 < ky < NY/2 counts twice (its Hermitian twin -ky has the same power and bins).
            // Task = (bin, chunk of rows): float64 sums in sample order, then the chunks of a bin in chunk order: bit-reproducible.
            const int nb = p.nbins;
            int rcn = T / nb;
            rcn = rcn < 1 ? 1 : (rcn > NROW + 1 ? NROW + 1 : rcn);
            double* part = reinterpret_cast<double*>(smem_raw + G::EFA);  // [rcn][nb]
            if (tid < nb * rcn) {
                const int bn = tid % nb, rc = tid / nb;
                double acc = 0.0;
                for (int ky = rc; ky <= NROW; ky += rcn) {
                    const unsigned short* __restrict__ fr = p.tfirst + (size_t)ky * (nb + 1) + bn;
                    const int s = fr[0], e = fr[1];
                    const float* r = Lf + ky * PF;
                    double rs = 0.0;
                    const int e1 = e < NX / 2 + 1 ? e : NX / 2 + 1;
                    for (int m = s; m < e1; ++m) rs += (double)r[m];                       // kx = m
                    const int ms = s > 1 ? s : 1, me = e < NX / 2 ? e : NX / 2;
                    for (int m = ms; m < me; ++m) rs += (double)r[NX - m];                 // kx = NX - m
                    acc += (ky != 0 && ky != NROW) ? 2.0 * rs : rs;
                }
                part[rc * nb + bn] = acc;
            }
            __syncthreads();
            if (tid < nb) {
                double tot = 0.0;
                for (int rc = 0; rc < rcn; ++rc) tot += part[rc * nb + tid];
                p.iso[(size_t)sl * nb + tid] = tot;
            }
            if (ISO == 2) return;  // (the next slab's first exchange starts with a barrier)
        }
        // ---- every output row whole: row ky (<= NY/2) rotated by the fftshift, row NY - ky reversed (F[-ky][-kx] = conj F[ky][kx])
        float* __restrict__ o = p.out + (size_t)sl * NY * NX;
        for (int e = tid; e < NY * NX / 4; e += T) {
            const int orow = e / (NX / 4), ch = e % (NX / 4);  // output row, float4 chunk
            const int ky = (orow - p.shift_y) & (NY - 1);      // unshifted frequency index of this output row
            const bool mir = ky > NY / 2;
            const float* r = Lf + (mir ? NY - ky : ky) * PF;
            const int kx0 = (4 * ch - p.shift_x) & (NX - 1);  // unshifted kx of the chunk's first sample
            F4 v;
            if (!mir) { v.x = r[kx0]; v.y = r[kx0 + 1]; v.z = r[kx0 + 2]; v.w = r[kx0 + 3]; }
            else { v.x = r[(NX - kx0) & (NX - 1)]; v.y = r[NX - kx0 - 1]; v.z = r[NX - kx0 - 2]; v.w = r[NX - kx0 - 3]; }
            xrft_store_nt(o + (size_t)orow * NX + 4 * ch, v);
        }
    };
    fastr_stagger(p.stagger);
    bool staged = false;
    long long prev = 0;
    for (long long slab = blockIdx.x;; slab += gridDim.x) {
        const bool have = slab < p.nslabs;
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);  // (nothing derived from the thread index is hoisted out of the slab loop and spilled: fastr.h)
        const int jp = tid % NXP, i0 = tid / NXP;  // packed column, first row
        cf a[32], b[32];
        if (have) {  // rows i0 + RY q, columns 2 jp, 2 jp + 1 (a uniform base + 32-bit offsets: no address pairs in registers)
            const char* __restrict__ base = reinterpret_cast<const char*>(p.in + (size_t)slab * NY * NX);
            const unsigned off = (unsigned)tid * 8u;
#pragma unroll
            for (int q = 0; q < 32; ++q) a[q] = *reinterpret_cast<const cf*>(base + (off + (unsigned)(q * T) * 8u));
        }
        if (PRE && staged) emit(prev);
        if (!have) break;
        if (p.detrend) {
            // S0 = sum x, Si = sum (i - ibar) x, Sj = sum (j - jbar) x over the slab, float64.  With u_q = x[i][2jp] + x[i][2jp+1]:
            //   Si = (i0 - ibar) sum u_q + RY sum q u_q,   Sj = (2 jp - jbar) sum u_q + sum x[i][2jp+1]
            constexpr double IBAR = 0.5 * (NY - 1), JBAR = 0.5 * (NX - 1);
            double U = 0.0, V = 0.0, I = 0.0;
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const double u = (double)a[q].re + (double)a[q].im;
                U += u;
                V = fma((double)q, u, V);
                I += (double)a[q].im;
            }
            double s0 = U, si = fma((double)i0 - IBAR, U, (double)RY * V), sj = fma((double)(2 * jp) - JBAR, U, I);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { s0 += __shfl_xor(s0, m); si += __shfl_xor(si, m); sj += __shfl_xor(sj, m); }
            if ((tid & 63) == 0) { red[(tid >> 6) * 3] = s0; red[(tid >> 6) * 3 + 1] = si; red[(tid >> 6) * 3 + 2] = sj; }
            __syncthreads();
            double t0 = 0.0, t1 = 0.0, t2 = 0.0;  // (`red` is next written a slab later, behind the barriers of the exchanges)
#pragma unroll
            for (int w = 0; w < G::NW; ++w) {
                t0 += red[3 * w]; t1 += red[3 * w + 1]; t2 += red[3 * w + 2];
                if ((w & 3) == 3) fastr_sched_fence();  // (four waves' sums at a time: all 48 values at once are 96 registers beside the slab's 64)
            }
            constexpr double INV_N2 = 1.0 / ((double)NY * NX);
            constexpr double INV_SI = 12.0 / ((double)NX * NY * ((double)NY * NY - 1.0)), INV_SJ = 12.0 / ((double)NY * NX * ((double)NX * NX - 1.0));
            const double c1 = p.detrend == 2 ? t1 * INV_SI : 0.0, c2 = p.detrend == 2 ? t2 * INV_SJ : 0.0;
            const double l0 = t0 * INV_N2 + c1 * ((double)i0 - IBAR) + c2 * ((double)(2 * jp) - JBAR), dl = (double)RY * c1;  // the plane at (i0 + RY q, 2 jp): l0 + dl q
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
                for (int q = 0; q < 16; ++q) wy[q] = p.win_y[i0 + RY * (16 * g + q)];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[16 * g + q] = mk<float>(a[16 * g + q].re * (wy[q] * wx.re), a[16 * g + q].im * (wy[q] * wx.im));
                fastr_sched_fence();
            }
        }
        // ---- y, stage 1: over q -> k1, x W_NY^(i0 k1)
        dft32f(a);
        twiddle32f(a, p.tw_y[i0]);
        // ---- exchange 1: writer (jp, i0) registers k1 -> reader (jp, c) registers ((g, w), i0): element of half h (i0 in [HY h, HY h + HY)) at
        // jp P1 + c 16 + (4 g + w) HY + (i0 - HY h).  P1 is odd: lane strides of 2 dwords (mod 32 and mod 64), conflict-free.
        const int c = i0;  // the reader's class: the thread set {jp + NXP m} serves a packed column before and after
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if (i0 / HY == h) {
                cf* dst = L + jp * P1 + (i0 % HY);
#pragma unroll
                for (int k1 = 0; k1 < 32; ++k1) dst[(fasts_c8(k1) / KGY) * 16 + (4 * (fasts_c8(k1) % KGY) + fasts_w(k1)) * HY] = a[k1];
            }
            __syncthreads();
            const cf* s = L + jp * P1 + c * 16;
#pragma unroll
            for (int gw = 0; gw < 4 * KGY; ++gw)
#pragma unroll
                for (int e = 0; e < HY; ++e) b[gw * RY + HY * h + e] = s[gw * HY + e];
        }
        // ---- y, stage 2: DFT_RY over i0 -> k2; then the packed columns are split in place
        fasts_dft_r<RY>(b);
        fasts_split<RY>(b, c);
        // ---- exchange 2: (jp, c) holds the rows {k1 + 32 k2, k2 < RY/2} of its k1 sets: columns 2 jp (lower registers) and 2 jp + 1 (their
        // partners) -> reader (row, x0) holding x = x0 + RX q'.  Half h carries the columns x in [NX/2 h, NX/2 h + NX/2): element (row, x) at
        // row PX + (x - NX/2 h); writers = the threads with jp in that half of the packed columns, readers all, q' in [16 h, 16 h + 16).
        const int row = tid / RX, x0 = tid % RX;  // after the exchange: tid = x0 + RX row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if (jp / (NXP / 2) == h) {
                cf* dst = L + 2 * (jp % (NXP / 2));
#pragma unroll
                for (int g = 0; g < KGY; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w)
#pragma unroll
                        for (int k2 = 0; k2 < HY; ++k2) {
                            // the even sequence's value at row idx (column 2 jp) and the odd one's (column 2 jp + 1)
                            // (two 8-byte stores: a 16-byte one wants its four registers adjacent -- copies beside a full register file)
                            const int r = fasts_k1(c * KGY + g, w) + 32 * k2;
                            dst[r * PX] = b[(4 * g + w) * RY + k2];
                            dst[r * PX + 1] = b[(4 * g + (w ^ 1)) * RY + RY - 1 - k2];
                        }
            }
            __syncthreads();
            const cf* s = L + row * PX + x0;
#pragma unroll
            for (int q = 0; q < 16; ++q) a[16 * h + q] = s[RX * q];
        }
        // ---- x, stage 1: over q' -> kx1, x W_NX^(x0 kx1)
        dft32f(a);
        twiddle32f(a, p.tw_x[x0]);
        // ---- exchange 3: within the RX adjacent lanes of a row: writer (row, x0) registers kx1 -> reader (row, cx) registers ((g, w), x0).
        // Element at (row - NROW/2 h) P3 + cx 33 + (4 g + w) RX + x0; the rows of half h are the threads [T/2 h, T/2 h + T/2).
        const int cx = x0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if (tid / (T / 2) == h) {
                cf* dst = L + (row % (NROW / 2)) * P3 + x0;
#pragma unroll
                for (int k1 = 0; k1 < 32; ++k1) dst[(fasts_c8(k1) / KGX) * 33 + (4 * (fasts_c8(k1) % KGX) + fasts_w(k1)) * RX] = a[k1];
            }
            __syncthreads();
            if (tid / (T / 2) == h) {  // (into `a` itself: a thread writes and reads in ONE half, its old values are dead when the new ones arrive)
                const cf* s = L + (row % (NROW / 2)) * P3 + cx * 33;
#pragma unroll
                for (int e = 0; e < 32; ++e) a[e] = s[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) b[e] = a[e];
        // ---- x, stage 2: DFT_RX over x0 -> kx2: b[(4 g + w) RX + kx2] = F[row][kx1 + 32 kx2]; the packed row 0 is split (rows 0 and NY/2)
        fasts_dft_r<RX>(b);
        if (row == 0) fasts_split<RX>(b, cx);
        if (MODE == 0) {
            // ---- complex output: F scale staged as complex rows, half of the rows at a time (rows [NROW/2 h, NROW/2 h + NROW/2) in slots
            // 0 .. NROW/2 - 1; the Nyquist row NY/2, which the threads of row 0 hold, in slot NROW/2 of half 0), then every output row whole:
            // row ky rotated by the fftshift, row NY - ky reversed and conjugated (F[-ky][-kx] = conj F[ky][kx]), x the phase factors
            cf* __restrict__ oc = reinterpret_cast<cf*>(p.out) + (size_t)slab * NY * NX;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();
                if (row / (NROW / 2) == h) {
                    if (row != 0) {
                        cf* dst = L + (row % (NROW / 2)) * PF;
#pragma unroll
                        for (int g = 0; g < KGX; ++g)
#pragma unroll
                            for (int w = 0; w < 4; ++w)
#pragma unroll
                                for (int k2 = 0; k2 < RX; ++k2) dst[fasts_k1(cx * KGX + g, w) + 32 * k2] = cscale(b[(4 * g + w) * RX + k2], p.scale);
                    } else {
#pragma unroll
                        for (int g = 0; g < KGX; ++g)
#pragma unroll
                            for (int w = 0; w < 4; ++w)
#pragma unroll
                                for (int k2 = 0; k2 < HX; ++k2) {
                                    const int idx = fasts_k1(cx * KGX + g, w) + 32 * k2;
                                    const cf e = cscale(b[(4 * g + w) * RX + k2], p.scale), o = cscale(b[(4 * g + (w ^ 1)) * RX + RX - 1 - k2], p.scale);
                                    cf* r0 = L;                     // row 0
                                    cf* rn = L + (NROW / 2) * PF;   // row NY/2
                                    if (idx == 0) {  // e = F[0][0] + i F[0][NX/2], o = F[NY/2][0] + i F[NY/2][NX/2] (four real samples)
                                        r0[0] = mk<float>(e.re, 0.f); r0[NX / 2] = mk<float>(e.im, 0.f);
                                        rn[0] = mk<float>(o.re, 0.f); rn[NX / 2] = mk<float>(o.im, 0.f);
                                    } else {         // rows 0 and NY/2 of a real field's spectrum are Hermitian in kx
                                        r0[idx] = e; r0[NX - idx] = cconj(e);
                                        rn[idx] = o; rn[NX - idx] = cconj(o);
                                    }
                                }
                    }
                }
                __syncthreads();
                constexpr int NSL = NROW / 2 + 1;  // slots (the last one only in half 0)
                for (int e = tid; e < NSL * 2 * (NX / 2); e += T) {
                    const int ch = e % (NX / 2), rr = e / (NX / 2), sl = rr >> 1, mir = rr & 1;
                    const int ky = sl == NROW / 2 ? NROW : sl + h * (NROW / 2);
                    if ((sl == NROW / 2 && h == 1) || (mir && (ky == 0 || ky == NROW))) continue;
                    const cf* r = L + sl * PF;
                    const int c0 = 2 * ch;                                              // output columns c0, c0 + 1
                    const int fx0 = (c0 - p.shift_x) & (NX - 1), fx1 = (c0 + 1 - p.shift_x) & (NX - 1);  // their unshifted frequency indices
                    const int fy = mir ? NY - ky : ky;
                    cf v0, v1;
                    if (!mir) { v0 = r[fx0]; v1 = r[fx1]; }
                    else { v0 = cconj(r[(NX - fx0) & (NX - 1)]); v1 = cconj(r[(NX - fx1) & (NX - 1)]); }
                    if (p.ph_on) {
                        const cf py = p.ph_y[fy];
                        v0 = cmul(v0, cmul(py, p.ph_x[fx0]));
                        v1 = cmul(v1, cmul(py, p.ph_x[fx1]));
                    }
                    xrft_store_nt2(oc + (size_t)((fy + p.shift_y) & (NY - 1)) * NX + c0, v0, v1);
                }
            }
            continue;
        }
        // ---- |F|^2 scale, staged as float rows: row r (0 <= r <= NY/2) at r PF + kx
        __syncthreads();
        if (row != 0) {
#pragma unroll
            for (int g = 0; g < KGX; ++g)
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int k2 = 0; k2 < RX; ++k2) {
                        const cf v = b[(4 * g + w) * RX + k2];
                        Lf[row * PF + fasts_k1(cx * KGX + g, w) + 32 * k2] = (v.re * v.re + v.im * v.im) * p.scale;
                    }
        } else {
            // the packed row was split: for set g, member w, k2 < RX/2 the lower register holds F[0][idx], its partner F[NY/2][idx],
            // idx = kx1 + 32 k2 (|F[.][NX - idx]| = |F[.][idx]|: both samples are written); idx = 0 carries the four real corner samples
#pragma unroll
            for (int g = 0; g < KGX; ++g)
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int k2 = 0; k2 < HX; ++k2) {
                        const int idx = fasts_k1(cx * KGX + g, w) + 32 * k2;
                        const cf e = b[(4 * g + w) * RX + k2], o = b[(4 * g + (w ^ 1)) * RX + RX - 1 - k2];
                        if (idx == 0) {  // e = F[0][0] + i F[0][NX/2], o = F[NY/2][0] + i F[NY/2][NX/2]
                            Lf[0] = e.re * e.re * p.scale; Lf[NX / 2] = e.im * e.im * p.scale;
                            Lf[NROW * PF] = o.re * o.re * p.scale; Lf[NROW * PF + NX / 2] = o.im * o.im * p.scale;
                        } else {
                            const float pe = (e.re * e.re + e.im * e.im) * p.scale, po = (o.re * o.re + o.im * o.im) * p.scale;
                            Lf[idx] = pe; Lf[NX - idx] = pe;
                            Lf[NROW * PF + idx] = po; Lf[NROW * PF + NX - idx] = po;
                        }
                    }
        }
        __syncthreads();
        staged = true;  // (MODE 1: emitted at the top of the next trip)
        prev = slab;
    }
}

}  // namespace xrft

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

// 16-byte store that is not kept in L2 / the Infinity Cache
__device__ __forceinline__ void xrft_store_nt(float* dst, F4 v) {
#ifdef XRFT_EMULATE
    *reinterpret_cast<F4*>(dst) = v;
#else
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(dst));
#endif
}

// FastY::tune (XRFTHIP_YTUNE, read once at plan creation): cache policies of the four streams and a start stagger, kept as knobs
// because the right setting is a property of the memory system, measured (scripts/tune_yf.py -> profiles/r03_tune_yf.txt):
//   bits 0-1  pass 1, stores of the intermediate: 0 non-temporal, 1 plain, 2 write-through (sc1)
//   bit  2    pass 1, loads of the input non-temporal
//   bit  3    pass 2, loads of the intermediate non-temporal
//   bit  4    pass 2, stores of the result plain instead of non-temporal
//   bit  21   pass 1: the workgroups that share the input's 128-byte lines (consecutive column blocks of one XCD) meet on a counter
//             before they load, with a time-out (round 4: does the 1.45x input over-fetch go away when the sharers load together?)
//   bits 8-15 the second workgroup of every CU (blocks 256..511 of a launch) starts n x 3.4 us late: the two residents of a CU
//             then run their load / transform / store phases out of step for the whole launch
// The product build compiles the default in (a run-time policy in the store loops costs the column kernel a spilled register);
// scripts/build_tune_yf.sh builds a second library with -DXRFT_YTUNE_RT whose kernels read FastY::tune, for scripts/tune_yf.py.
constexpr long long kYTuneDefault = 0;
#ifdef XRFT_YTUNE_RT
#define XRFT_YTUNE(p) ((p).tune)
#else
#define XRFT_YTUNE(p) ((int)kYTuneDefault)
#endif

// 16-byte store at base + off with a run-time cache policy: 0 non-temporal, 1 plain, 2 write-through to memory (sc1; `base` must be
// wave-uniform: it becomes the buffer descriptor -- a per-lane base is a 64-iteration waterfall loop, 154 us per slab)
__device__ __forceinline__ void xrft_store_pol(char* base, unsigned off, F4 v, int pol) {
#ifdef XRFT_EMULATE
    *reinterpret_cast<F4*>(base + off) = v;
#else
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = {v.x, v.y, v.z, v.w};
    if (pol == 0) __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(base + off));
    else if (pol == 1) *reinterpret_cast<v4f*>(base + off) = t;
    else __builtin_amdgcn_raw_buffer_store_b128(t, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000), (int)off, 0, 16);
#endif
}
__device__ __forceinline__ F4 xrft_load_pol(const char* src, bool nt) {
#ifdef XRFT_EMULATE
    return *reinterpret_cast<const F4*>(src);
#else
    typedef float v4f __attribute__((ext_vector_type(4)));
    if (!nt) return *reinterpret_cast<const F4*>(src);
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src));
    F4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#endif
}
__device__ __forceinline__ cf xrft_load8_pol(const char* src, bool nt) {
#ifdef XRFT_EMULATE
    return *reinterpret_cast<const cf*>(src);
#else
    typedef float v2f __attribute__((ext_vector_type(2)));
    if (!nt) return *reinterpret_cast<const cf*>(src);
    const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(src));
    return mk<float>(t.x, t.y);
#endif
}
// start stagger (FastY::tune bits 8-15): blocks 256..511 = the second resident of every CU at launch start
__device__ __forceinline__ void xrft_stagger(int tune) {
#ifndef XRFT_EMULATE
    const int n = (tune >> 8) & 0xff;
    if (n && blockIdx.x >= 256u && blockIdx.x < 512u)
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
#endif
}

struct FastY {
    const float* in;         // [slab][ny][nx] float32
    cf* w2;                  // intermediate, see above
    const cf* w2b;           // cross spectra: the intermediate of field 1 (pass 2 only)
    void* out;               // [slab][ny][nx]: float32 power / phase, or complex64 (may be null with ISO)
    const cf* tw_x;          // W_nx^k
    const cf* tw_y;          // W_ny^k
    const float* win_y;      // never null (ones when there is no window)
    const float* win_x;
    const float* win2d;      // four-step 1-D with a window: w[n] laid out like the slab, [ny][nx] (else unused)
    double* colfit;          // [slab][nx][4]: per column sum r, sum (i - ibar) r of the residual r = d - line, and the line pass 1 subtracted as (value at ibar, slope)
    const float* corr;       // [slab][nx][2]: wx[x] * (subtracted line - plane fit) as (offset at ibar, slope), from fasty_fit_kernel
    const float* corr_b;     // ... of field 1 (cross spectra)
    const cf* ph_y;          // complex modes: true-phase factor per unshifted ky (times (-1)^ky for an ifftshifted input), never null
    const cf* ph_x;          // (four-step 1-D: one table over the whole sequence, indexed by the unshifted sample index)
    const cf* tw_big;        // four-step 1-D: W_N^j, j < N / 2
    int ph_on;               // 0: every phase factor is 1 (true_phase off, no ifftshift): the tables are not read
    int half;                // real_dim: only kx = 0..nx/2 is stored, rows of nx/2 + 1 samples, unshifted (xrft.py:400-404)
    int realdim2;            // ... and 0 < kx < nx/2 counts twice (xrft.py:673-682)
    const cf* what0;         // FFT_y(wy)[ky], ky < nrow_pad (zero beyond ny/2)
    const cf* what1;         // FFT_y(wy * (i - (ny-1)/2))[ky]
    const unsigned* tcodes;  // radial bins [ky < nrow_pad][kx] in natural order: (direct + 1) | (mirror + 1) << 16 (0: not binned);
                             // tcodes_compact: [ky][kx / 16]: (first sample's bin + 1) | step mask << 16 (see fasty_rows_kernel)
    int tcodes_compact;
    const unsigned* twin;          // radial map: [unit] first bin | (last bin + 1) << 16 of the unit's rows: bins outside it are neither gathered, written nor reduced
    const unsigned short* tfirst;  // radial map: [ky < nrow_pad][nbins + 1], the smallest |kx| of a row whose bin is >= b (null: any map, the atomic tables)
    double* iso;             // [slab][nbins] per-bin sums (ISO)
    double* iso_part;        // [slab][row workgroup][nbins (x2 complex)]: per-workgroup partial sums, reduced in order
    int nbins;
    int ny, nx;
    int nrow_pad;            // rows of W2 per slab
    int l_cw, l_rk, l_2gy;   // log2 of CW, RK, 2 GY (layout of W2, fixed by ny)
    int detrend;             // 0 none, 1 constant, 2 linear
    int nslab;
    int shift_y, shift_x;    // 0 or n/2
    float scale;
    int tune;                // see kYTuneDefault
    long long* tim;          // fasty_isorows_kernel<.., TIM>: [workgroup][8] shader-clock sums of its phases (profiling build only; null otherwise)
    unsigned* rdv;           // tune bit 21: one arrival counter per (slab, group of the column blocks that share the input's 128-byte lines), zeroed per launch
};

// phase-ablation bits for profiling builds (scripts/gpu_ablate_yf.sh compiles variants with -DXRFT_YDBG=bits); 0 in the product
#ifndef XRFT_YDBG
#define XRFT_YDBG 0
#endif

// geometry of pass 1 for NY-point columns
template <int NY> struct YCols {
#ifdef XRFT_YCOLS_WIDE  // (round-6 experiment, scripts/prof.py cols-wide: a 1024-thread workgroup owns 16 columns of a 4096-point slab -- 64-byte row segments, two sharers per input line instead of four)
    static constexpr int THR = NY >= 4096 ? 1024 : NY >= 2048 ? 512 : 256;
#else
    static constexpr int THR = NY >= 2048 ? 512 : 256;
#endif
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
// (fft_p2_pair_w: the first-stage twiddle W_N^u handed in -- a persistent workgroup loads it once)
template <int N> __device__ __forceinline__ void fft_p2_pair_w(cf* a, cf* b, int u, cf* lds, const cf w1, const cf* tw2) {
    typedef P2<N> G;
    const int k1 = u / G::R3, v = u % G::R3;
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

template <int N> __device__ __forceinline__ void fft_p2_pair(cf* a, cf* b, int u, cf* lds, const cf* __restrict__ tw, const cf* tw2) {
    fft_p2_pair_w<N>(a, b, u, lds, tw[u], tw2);  // W_N^u
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
// W2D (four-step 1-D with a window): the window of a long sequence is not separable over its [ny][nx] view, so it comes from a
// table laid out like the slab, read at the samples' own offsets (w[nx i1 + i2]; shared by every slab: L2-resident).
template <int NY, bool DET, bool W2D = false>
__global__ void __launch_bounds__(YCols<NY>::THR, (YCols<NY>::THR >= 512 ? 4 : YCols<NY>::THR / 128)) fasty_cols_kernel(FastY p) {
    typedef P2<NY> G;
    typedef YCols<NY> Y;
    constexpr int NT = G::NT, GY = Y::GY, THR = Y::THR, GSTR = YLds<NY, GY>::GSTR;
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GY, u = tid / GY;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GY * GSTR;
    xrft_stagger(XRFT_YTUNE(p));
    fill_tw2<NY>(tw2, p.tw_y, tid, THR);
    // unit = (slab, column block).  Blocks b, b+8, b+16, ... run on one XCD (round-robin dispatch): give each XCD a
    // contiguous range of column blocks, so that the workgroups sharing a 128-byte line of the input share an L2.
    const int nxb = p.nx / Y::CW;
    int slab, xb;
    if ((nxb & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nxb >> 3;
        slab = j / per;
        int jj = j % per;
        // (tuning build: where the four sharers of a line go.  Workgroups k and k + 32 of an XCD land on one CU in the first round of
        // a launch -- scripts/ubench/cuid.hip -- so sharers 16 or 32 apart share an L1 while they stay in step)
        const int mp = (XRFT_YTUNE(p) >> 19) & 3;
        if (mp == 1 && (per & 3) == 0) jj = (jj % (per >> 2)) * 4 + jj / (per >> 2);
        else if (mp == 2 && per == 64) jj = ((jj & 31) >> 1) * 4 + (jj & 1) * 2 + (jj >> 5);
        xb = xcd * per + jj;
    } else {
        slab = blockIdx.x / nxb;
        xb = blockIdx.x % nxb;
    }
    if ((XRFT_YTUNE(p) >> 21) & 1) {
#ifndef XRFT_EMULATE
        // rendezvous of the sharers of a line group: 128 / (4 CW) column blocks, consecutive on one XCD (dispatched in order: at most
        // one group is partially resident, so the wait cannot deadlock; the time-out is a belt)
        constexpr int SH = 128 / (4 * Y::CW) > 1 ? 128 / (4 * Y::CW) : 1;
        if (SH > 1) {
            if (tid == 0) {
                unsigned* ctr = p.rdv + (size_t)slab * (nxb / SH) + xb / SH;
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int it = 0; it < 4000; ++it) {
                    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)SH) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
        }
#endif
    }
    const int x0 = xb * Y::CW + 4 * g;
    // uniform 64-bit base + one 32-bit per-lane byte offset (a slab is < 4 GB): scalar-base loads, no 64-bit address per row
    const char* __restrict__ src = reinterpret_cast<const char*>(p.in + (size_t)slab * NY * p.nx + (size_t)xb * Y::CW);
    const unsigned off0 = ((unsigned)u * (unsigned)p.nx + 4u * (unsigned)g) * 4u, rstep = (unsigned)NT * (unsigned)p.nx * 4u;
    F4 wx = {1.f, 1.f, 1.f, 1.f};
    if (!W2D) wx = *reinterpret_cast<const F4*>(p.win_x + x0);
    const char* __restrict__ wsrc = reinterpret_cast<const char*>(p.win2d + (size_t)xb * Y::CW);
    // ---- detrend, fused, nothing on the critical path.  The plane of xrft/detrend.py:100-113 needs sums over the whole
    // slab, which exist only after this pass.  What is subtracted HERE, in y-space, only has to take the bulk of the trend
    // out (so that nothing cancels catastrophically in float32) and to be one line T + S i per column: it is an estimate
    // from a few reference rows of the column (below), which every thread of the group loads itself (the same
    // addresses in all lanes of a wave: one transaction) BEFORE its own 16 rows, so that the constants are ready when the
    // rows arrive.  Pass 2 adds the difference to the true plane back in the spectral domain,
    // wx[x] (alpha_x What0[ky] + gamma_x What1[ky]) with What0 = FFT(wy), What1 = FFT(wy (i - ibar)), from the exact column sums
    // (sum d, sum (i - ibar) d) that this kernel produces on the side: float32 over a thread's 16 rows, float64 wave
    // shuffles, one small LDS table that is summed AFTER the transforms -- no barrier of its own.
    // (A three-level LDS reduction in front of the transforms, with the rows held in registers meanwhile, cost 7 of 36 us
    // per slab, most of it through the 17 VGPRs it made the kernel spill: 36 MB of scratch traffic per slab.)
    constexpr double IBAR = 0.5 * (NY - 1);
    // reference rows: KREF ADJACENT rows around ny/4 and around 3 ny/4, the per-column MEDIAN of each triple.  Away from the edges
    // (where real fields carry their artefacts: coast lines, padding, tapering) and robust to a spike in any single row: what the
    // estimate misses costs float32 digits in the ky = 0, +-1 bins, where pass 2 has to cancel it (round 2 took the mean of rows
    // 0, 1 and ny-2, ny-1: 1e3-spikes there left 4e-3 relative errors in bins 1e-6 of the peak, tests/cases.py adversarial fields)
    constexpr int KREF = 3, NW = THR / 64, ITOP = NY / 4, IBOT = 3 * NY / 4;
    double* part = reinterpret_cast<double*>(tw2 + 16 * G::R3);  // [wave][g][8]
    float T[4] = {0.f, 0.f, 0.f, 0.f}, S[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
    F4 rt[KREF], rb[KREF];
    if (DET) {
        const unsigned offg = 16u * (unsigned)g, rowb = (unsigned)p.nx * 4u;
#pragma unroll
        for (int k = 0; k < KREF; ++k) {
            rt[k] = *reinterpret_cast<const F4*>(src + (offg + rowb * (unsigned)(ITOP - 1 + k)));
            rb[k] = *reinterpret_cast<const F4*>(src + (offg + rowb * (unsigned)(IBOT - 1 + k)));
        }
    }
    F4 raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) raw[q] = xrft_load_pol(src + (off0 + rstep * (unsigned)q), (XRFT_YTUNE(p) & 4) != 0);
    if (DET) {
        auto med3 = [](float x, float y, float z) { return fmaxf(fminf(x, y), fminf(fmaxf(x, y), z)); };
        const float mt[4] = {med3(rt[0].x, rt[1].x, rt[2].x), med3(rt[0].y, rt[1].y, rt[2].y), med3(rt[0].z, rt[1].z, rt[2].z), med3(rt[0].w, rt[1].w, rt[2].w)};
        const float mb[4] = {med3(rb[0].x, rb[1].x, rb[2].x), med3(rb[0].y, rb[1].y, rb[2].y), med3(rb[0].z, rb[1].z, rb[2].z), med3(rb[0].w, rb[1].w, rb[2].w)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float top = mt[c], bot = mb[c];  // the column near i = ITOP and i = IBOT
            const float Se = p.detrend == 2 ? (bot - top) * (1.0f / (IBOT - ITOP)) : 0.f;
            const float Te = p.detrend == 2 ? top - Se * (float)ITOP : 0.5f * (top + bot);  // value of the line at i = 0
            // The line is OUR choice (pass 2 corrects whatever is subtracted here), so both coefficients are rounded to one
            // coarse power-of-two grid 2^(e-20), 2^e <= |T| + |S| ny < 2^(e+1) (adding and subtracting C = 1.5 * 2^(e+3)
            // rounds to that grid): T + S i is then exact in float32 for every row, and x - (T + S i) has a single rounding,
            // relative to the already noise-sized result (a plain float32 evaluation of the trend leaves 6e-4 of max in the
            // ky = 0 bins, cf. fastp2_rows_kernel).  The slope keeps >= 9 bits: the residual line stays < 0.1 % of the trend.
            const float mag = fabsf(Te) + fabsf(Se) * (float)NY;
            const float C = __uint_as_float((__float_as_uint(mag) & 0x7f800000u) + (3u << 23)) * 1.5f;
            T[c] = (Te + C) - C; S[c] = (Se + C) - C;
            // The grid leaves the slope 8-9 bits: a residual line of up to 0.4 % of the trend (16 x the noise under a trend 1e4 x
            // the noise at ny = 4096, which cost the ky = +-1 bins 2.5e-3 of relative error).  What the grid dropped of the slope,
            // S2 = Se - S (exact: the two are within a factor two), leaves in a fused multiply-add, (x - L1(i)) - S2 i: the product
            // is exact inside the fma and the one rounding is at the size of the result, so the line subtracted is exactly
            // T + (S + S2) i with the slope at full precision (the offset's grid error is 2^-21 of the trend: nothing).
            S2[c] = Se - S[c];
        }
        if (u == 0) {  // what is subtracted, as (offset at ibar, slope)
            double* cfp = p.colfit + ((size_t)slab * p.nx + x0) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                cfp[4 * c + 2] = (double)T[c] + ((double)S[c] + (double)S2[c]) * IBAR;
                cfp[4 * c + 3] = (double)S[c] + (double)S2[c];
            }
        }
    }
    cf a[16], b[16];
    // column sums of the RESIDUAL d - (T + S i), float32 over this thread's rows: the residual is noise-sized, so nothing is lost
    // (sums of the raw samples lose the mean of a column to rounding once an offset or trend is >~ 1e4 times the signal: with
    // 1e6 + N(0, 1) the 16-row partial sums sit at 2^24 and the ky = 0 row came out 90 % wrong); fasty_fit_kernel adds the
    // subtracted line back in float64
    float f0[4] = {0.f, 0.f, 0.f, 0.f}, f1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float fi = (float)(u + NT * q);
        float wy = 1.f;
        if (!W2D) wy = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.win_y) + (unsigned)(u + NT * q) * 4u);
        const float v0 = DET ? fmaf(-S2[0], fi, raw[q].x - fmaf(S[0], fi, T[0])) : raw[q].x;
        const float v1 = DET ? fmaf(-S2[1], fi, raw[q].y - fmaf(S[1], fi, T[1])) : raw[q].y;
        const float v2 = DET ? fmaf(-S2[2], fi, raw[q].z - fmaf(S[2], fi, T[2])) : raw[q].z;
        const float v3 = DET ? fmaf(-S2[3], fi, raw[q].w - fmaf(S[3], fi, T[3])) : raw[q].w;
        if (DET) {
            const float fq = (float)q;
            f0[0] += v0; f1[0] = fmaf(fq, v0, f1[0]);
            f0[1] += v1; f1[1] = fmaf(fq, v1, f1[1]);
            f0[2] += v2; f1[2] = fmaf(fq, v2, f1[2]);
            f0[3] += v3; f1[3] = fmaf(fq, v3, f1[3]);
        }
        if (W2D) {
            if ((q & 3) == 0) asm volatile("" ::: "memory");  // window loads in batches of four rows (all sixteen hoisted: 40 spilled registers at 4096)
            const F4 ww = *reinterpret_cast<const F4*>(wsrc + (off0 + rstep * (unsigned)q));
            a[q] = mk<float>(v0 * ww.x, v1 * ww.y);
            b[q] = mk<float>(v2 * ww.z, v3 * ww.w);
        } else {
            a[q] = mk<float>(v0 * (wy * wx.x), v1 * (wy * wx.y));
            b[q] = mk<float>(v2 * (wy * wx.z), v3 * (wy * wx.w));
        }
    }
    if (DET) {
        // sum (i - ibar) r over i = u + NT q is (u - ibar) S0 + NT sum q r.  float32 through the wave (the sums are of noise-sized
        // residuals: 1e-7 of the noise is lost; as float64 -- needed when these were sums of the raw samples -- the eight values
        // were sixteen registers beside the two transforms' sixty-four and spilled), float64 from the per-wave table on
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            v[c] = f0[c];
            v[4 + c] = fmaf((float)u - (float)IBAR, f0[c], (float)NT * f1[c]);
        }
#pragma unroll
        for (int m = GY; m < 64; m <<= 1)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += __shfl_xor(v[c], m);
        if ((tid & 63) < GY) {  // the lane with the wave's first u: lane = g
#pragma unroll
            for (int c = 0; c < 8; ++c) part[((tid >> 6) * GY + g) * 8 + c] = (double)v[c];
        }
    }
    if (!(XRFT_YDBG & 32)) fft_p2_pair<NY>(a, b, u, mine, p.tw_y, tw2);
    // (the lane indices are re-derived from an opaque copy of the thread index: carried from the top of the kernel through the
    // transforms they were spilled -- 3 dwords per lane = 3 MB of scratch traffic per 4096^2 slab each way)
    int tid2 = threadIdx.x;
    XRFT_OPAQUE(tid2);
    if (DET && tid2 < 8 * GY) {  // the transforms' barriers have made every wave's partial sums visible
        const int c = tid2 / GY, gg = tid2 % GY;
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += part[(w * GY + gg) * 8 + c];
        p.colfit[((size_t)slab * p.nx + xb * Y::CW + 4 * gg + (c & 3)) * 4 + (c >> 2)] = acc;
    }
    // split the packed transforms: Ra[k] = (Z[k] + conj Z[N-k]) / 2, Rb[k] = (Z[k] - conj Z[N-k]) / (2i), k = u + NT q (q < 8)
    // and k = NY/2; (Ra, Rb) = two adjacent columns = one 16-byte store; lanes (u..u+RK-1, all g) complete a line
    char* __restrict__ w2s = reinterpret_cast<char*>(p.w2 + (size_t)slab * p.nrow_pad * p.nx);
    const int g2 = tid2 % GY, u2 = tid2 / GY;
    cf* mine2 = lds + g2 * GSTR;
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const cf* z = set == 0 ? a : b;
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) mine2[nat16(held_k<NY>(u2, bb, k3))] = z[bb * G::R3 + k3];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int k = u2 + NT * q;
            if (q < 8 || u2 == 0) {
                const cf zk = mine2[nat16(k & (NY - 1))];
                const cf zc = cconj(mine2[nat16((NY - k) & (NY - 1))]);
                const cf ra = cscale(zk + zc, 0.5f), rb = cscale(mul_mi(zk - zc), 0.5f);
                const unsigned off = ((((unsigned)(k / Y::RK) * (unsigned)nxb + (unsigned)xb) * 2u + set) * Y::LBS) + (k % Y::RK) * (2 * GY) + 2 * g2;
                F4 o; o.x = ra.re; o.y = ra.im; o.z = rb.re; o.w = rb.im;
                // non-temporal: the next reader is another kernel, a whole group of slabs later; kept out of L2 the lines leave
                // it to the input, whose 128-byte lines are shared by four workgroups (PMC: 1.34x over-fetch with plain stores)
                if (!(XRFT_YDBG & 64) || o.x == 1.2345f) xrft_store_pol(w2s, off * 8u, o, XRFT_YTUNE(p) & 3);
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
// Small workgroups -- 128 threads up to 1024-point rows, 256 at 2048: twice as many per CU as with 256 / 512 -- interleave their
// load / transform / store phases better: row pass -19 % at 256 (PS (4096, 256, 256) 256 -> 286 GFFT/s), -6 ... -8 % at 512, 1024,
// 2048.  Not at 4096 (one sequence pair per 256-thread workgroup: 24.1 -> 32.5 us), and not in the four-step form, whose
// transposed runs are one sample per row of the unit: 32 rows = a whole line (dft (1024, 65536): 170 vs 156 GFFT/s).
template <int NX, bool FS = false> struct YRows {
    static constexpr int THR = NX >= 4096 ? 512 : NX >= 2048 ? 256 : ((NX == 256 && FS) ? 256 : 128);
    static constexpr int NT = NX / 16;
    static constexpr int GX = THR / NT;  // sequences (pairs of rows) in lockstep: 8, 4, 2, 2, 2 (four-step: 16)
    static constexpr int RPU = 2 * GX;   // rows per workgroup
    static constexpr int RS = NX + NX / 16;  // floats per staged row (nat16 padding)
};

// element offset of (ky, x) inside one slab of W2 (< 2^24 elements)
__device__ __forceinline__ unsigned w2_offset(const FastY& p, int ky, int x) {
    const unsigned nxb = (unsigned)p.nx >> p.l_cw;
    const unsigned blk = (((unsigned)ky >> p.l_rk) * nxb + ((unsigned)x >> p.l_cw)) * 2u + (((unsigned)x >> 1) & 1u);
    return (blk << (p.l_rk + p.l_2gy)) + (((unsigned)ky & ((1u << p.l_rk) - 1u)) << p.l_2gy) + ((((unsigned)x & ((1u << p.l_cw) - 1u)) >> 2) << 1) + ((unsigned)x & 1u);
}

// float -> int64 fixed point with 40 fractional bits below 2^(eb - 127), eb = biased exponent of a bound |v| < 2^(eb - 126):
// integer arithmetic only (no float64 <-> int64 conversions, which gfx950 emulates); rounds to nearest at 2^-40 of the bound
__device__ __forceinline__ long long fixed40(float v, int eb) {
    const unsigned bits = __float_as_uint(v);
    const int ex = (int)((bits >> 23) & 0xffu);
    if (ex == 0) return 0;  // zero / denormal
    long long m = (long long)((bits & 0x7fffffu) | 0x800000u);  // 24-bit mantissa: |v| = m 2^(ex - 150)
    const int sh = ex - eb + 17;                                // q = |v| 2^(40 + 127 - eb) = m 2^(ex - eb + 17), sh <= 17
    m = sh >= 0 ? (m << sh) : (sh > -25 ? ((m + (1ll << (-sh - 1))) >> (-sh)) : 0);  // (rounds to nearest; below half a unit: 0)
    return (bits >> 31) ? -m : m;
}

// 16-byte non-temporal store of two complex samples
__device__ __forceinline__ void xrft_store_nt2(cf* dst, cf v0, cf v1) {
    F4 o; o.x = v0.re; o.y = v0.im; o.z = v1.re; o.w = v1.im;
    xrft_store_nt(reinterpret_cast<float*>(dst), o);
}

// an unaligned 32-bit read of two adjacent 16-bit table entries (one global_load_dword on gfx950)
struct __attribute__((packed, aligned(2))) U16Pair { unsigned v; };


// Sum of the NRW (2 .. 16, a power of two) adjacent lanes that hold the rows of one bin, valid in the FIRST lane of the group, in the fixed
// tree order ((r0 + r1) + (r2 + r3)) + ... -- what `v += __shfl_down(v, 1, NRW); v += __shfl_down(v, 2, NRW); ...` yields there.
// DPP row shifts (lane i reads lane i + m of its row of 16): no trip through the LDS crossbar.
template <int NRW> __device__ __forceinline__ double quad_rows_sum(double v) {
    static_assert(NRW == 2 || NRW == 4 || NRW == 8 || NRW == 16, "the rows of a bin are adjacent lanes of one DPP row");
#ifdef XRFT_EMULATE
#pragma unroll
    for (int m = 1; m < NRW; m <<= 1) v += __shfl_down(v, m, NRW);
    return v;
#else
    auto from = [](double x, auto ctrl) -> double {  // (lanes whose source falls off the row read 0: never the group's first lane)
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), decltype(ctrl)::value, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, false);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    v += from(v, std::integral_constant<int, 0x101>());                 // row_shl:1
    if (NRW >= 4) v += from(v, std::integral_constant<int, 0x102>());   // row_shl:2
    if (NRW >= 8) v += from(v, std::integral_constant<int, 0x104>());   // row_shl:4
    if (NRW >= 16) v += from(v, std::integral_constant<int, 0x108>());  // row_shl:8
    return v;
#endif
}

// ------------------------------------------------------------------------------------------------
// The per-bin gather of the radial sums (xrft.py:895-906) from rows staged in LDS, shared by fasty_rows_kernel<.., ISO> and
// fasty_isorows_kernel (fasty_iso.h).  Step (1) of those kernels left the sum of every run of equal bins of a 16-sample segment on
// the run's last sample; here the owner of a (bin, row) walks the bin's two kx ranges of that row, one staged value per segment it
// touches, and adds them in float64 in a fixed order (+kx side by rising segment, then the -kx side); the NRW rows of a bin are
// adjacent lanes of a quad and meet in lane order.  The phase is a chain of LDS latencies, not of work (measured with the profiling
// build of fasty_isorows_kernel: 7900 shader cycles per unit of four 4096-sample rows with a branch and a wait per read, 5600 so):
// the first two segments of either side of KB sweeps are read up front with UNCONDITIONAL, masked reads (a bin narrower than 16
// samples touches no more), the few lanes whose bin hugs |k| = ky walk on four reads per trip, and the rows of a bin are added by DPP
// quad permutes instead of trips through the LDS crossbar.
//   rowp: this lane's staged row; rg[k]: the |kx| range s | e << 16 of bin blo + (k0 + k) BPI + blane in this row, k < kn <= KB
// ------------------------------------------------------------------------------------------------
template <int NX, int CPS>
__device__ __forceinline__ void radial_walk(const float* rowp, int seg0, int lim, double& sre, double& sim) {
    constexpr bool TWO = CPS == 2;
    for (int seg = seg0; 16 * seg < lim; seg += 4) {
        float w[4][CPS];
        int wo[4];
        unsigned wk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = 16 * (seg + j) < lim;
            wo[j] = CPS * nat16(on ? min(lim, 16 * (seg + j + 1)) - 1 : 0);
            wk[j] = on ? 0xffffffffu : 0u;
            XRFT_OPAQUE(wk[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CPS; ++c) w[j][c] = rowp[wo[j] + c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CPS; ++c) w[j][c] = __uint_as_float(__float_as_uint(w[j][c]) & wk[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sre += (double)w[j][0];
            if (TWO) sim += (double)w[j][CPS - 1];
        }
    }
}

template <int NX, int NRW, int CPS, int BPI, int KB>
__device__ __forceinline__ void radial_gather_batch(const float* rowp, int row, bool live, bool twin, int blo, int bhi, int k0, int kn, const unsigned* rg,
                                                int blane, double* __restrict__ part, bool store) {
    constexpr bool TWO = CPS == 2;
    constexpr int HW = TWO ? 2 : 1;
    auto walk = [&](int seg0, int lim, double& sre, double& sim) { radial_walk<NX, CPS>(rowp, seg0, lim, sre, sim); };
    float pv[KB][4][CPS];
    int pa[KB][4];        // the walk beyond the two segments read up front: first segment / limit, either side
    int po[KB][4];        // float offsets of the four reads
    unsigned keep[KB][4];  // all ones: the read counts
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const int bn = blo + (k0 + k) * BPI + blane;
        const bool on = k < kn && live && bn < bhi;
        const int s_ = (int)(rg[k] & 0xffffu), e_ = (int)(rg[k] >> 16);
        const int e1 = min(e_, NX / 2);                               // +side: kx = s .. e1 - 1
        const int ms = max(s_, 1), me = min(e_, NX / 2 + 1);          // -side: kx = nx - |kx|, |kx| = ms .. me - 1
        const int lo = NX - (me - 1), hi1 = NX - ms + 1;
        const int sp = s_ >> 4, sm = lo >> 4;
        const bool vp0 = on && s_ < e1, vp1 = vp0 && 16 * (sp + 1) < e1;
        const bool vm0 = on && ms < me, vm1 = vm0 && 16 * (sm + 1) < hi1;
        const bool vv[4] = {vp0, vp1, vm0, vm1};
        const int aa[4] = {min(e1, 16 * (sp + 1)) - 1, min(e1, 16 * (sp + 2)) - 1, min(hi1, 16 * (sm + 1)) - 1, min(hi1, 16 * (sm + 2)) - 1};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // UNCONDITIONAL reads of valid slots, masked afterwards (a select on the loaded value is turned back into a branch around
            // the read: twelve serialised LDS round trips per batch; the opaque mask keeps the `and` from being folded into a select)
            po[k][j] = CPS * nat16(vv[j] ? aa[j] : 0);
            keep[k][j] = vv[j] ? 0xffffffffu : 0u;
            XRFT_OPAQUE(keep[k][j]);
        }
        pa[k][0] = vp1 ? sp + 2 : NX; pa[k][1] = e1; pa[k][2] = vm1 ? sm + 2 : NX; pa[k][3] = hi1;
    }
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CPS; ++c) pv[k][j][c] = rowp[po[k][j] + c];
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CPS; ++c) pv[k][j][c] = __uint_as_float(__float_as_uint(pv[k][j][c]) & keep[k][j]);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        if (k >= kn) continue;
        const int bn = blo + (k0 + k) * BPI + blane;
        double sre = 0.0, sim = 0.0;
        sre += (double)pv[k][0][0]; if (TWO) sim += (double)pv[k][0][CPS - 1];
        sre += (double)pv[k][1][0]; if (TWO) sim += (double)pv[k][1][CPS - 1];
        walk(pa[k][0], pa[k][1], sre, sim);
        sre += (double)pv[k][2][0]; if (TWO) sim += (double)pv[k][2][CPS - 1];
        sre += (double)pv[k][3][0]; if (TWO) sim += (double)pv[k][3][CPS - 1];
        walk(pa[k][2], pa[k][3], sre, sim);
        if (twin) { sre *= 2.0; sim = 0.0; }  // V + conj V (a power spectrum's two samples are equal)
        sre = quad_rows_sum<NRW>(sre);  // rows 0 .. NRW - 1 in a fixed tree order: (r0 + r1) + (r2 + r3)
        if (TWO) sim = quad_rows_sum<NRW>(sim);
        if (row == 0 && bn < bhi && store) {
            part[bn * HW] = sre;
            if (TWO) part[2 * bn + (TWO ? 1 : 0)] = sim;
        }
    }
}

// MODE = xrfthip_out_mode: 1 power (two rows of one field per thread), 0 complex (fft; the same, staged in two rounds),
// 2 cross / 3 cross phase (transform A = the row of field 0, transform B = the same row of field 1: F0 conj(F1) is formed in
// registers, so a cross spectrum costs ONE column pass per field and one row pass -- xrft.py:825).  The complex results
// carry the true-phase factors exp(-i 2 pi k lag) (xrft.py:462-469), indexed by unshifted frequency, applied in the store
// loop to direct and mirrored samples alike (an ifftshifted input is the sign (-1)^k folded into the tables).
//
// FS ("four-step"): the slab is ONE long real sequence of N = ny * nx samples, n = nx i1 + i2, and the two passes are the two
// steps of its 1-D transform X[k1 + ny k2] = sum_i2 W_N^(i2 k1) W_nx^(i2 k2) [ sum_i1 x[nx i1 + i2] W_ny^(i1 k1) ]: pass 1 is
// unchanged (columns = the inner sums, half spectrum k1 <= ny/2 of a real sequence), this pass multiplies row k1 by
// W_N^(i2 k1) before its transform and stores TRANSPOSED: the unit's rows k1 are consecutive output samples for a given k2
// (2 GX * 4 or GX * 8 bytes contiguous); the Hermitian mirror X[N - k] = conj X[k] is the reversed run.  (xrft.dft / fft /
// power_spectrum along one long axis, BASELINE.json configs[1]: 1-D (1024, 65536) float32.)
template <int NX, int MODE, bool ISO, bool FS = false, bool W2D = false>
__global__ void __launch_bounds__((YRows<NX, FS>::THR), ((ISO && YRows<NX, FS>::THR >= 256) ? 4 : YRows<NX, FS>::THR / 128 < 1 ? 1 : YRows<NX, FS>::THR / 128)) fasty_rows_kernel(FastY p) {
    static_assert(!W2D || FS, "the slab-shaped window belongs to the four-step form");
    static_assert(MODE == 1 || MODE == 2 || !ISO, "radial sums exist for power and cross spectra");
    static_assert(!FS || (MODE <= 1 && !ISO), "the four-step form serves fft and power_spectrum");
    typedef P2<NX> G;
    typedef YRows<NX, FS> R;
    constexpr bool TWO = MODE >= 2;  // two fields
    constexpr int NT = G::NT, GX = R::GX, THR = R::THR, RPU = TWO ? GX : 2 * GX, GSTR = YLds<NX, GX>::GSTR;
    constexpr int HW = MODE == 2 ? 2 : 1;  // doubles per radial bin
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* stg = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, g = tid % GX, u = tid / GX;
    cf* mine = lds + g * GSTR;
    cf* tw2 = lds + GX * GSTR;
    // (the radial-sum tables alias the transforms' LDS: a separate 12-20 KB would cost the second workgroup per CU)
    xrft_stagger(XRFT_YTUNE(p));
    fill_tw2<NX>(tw2, p.tw_x, tid, THR);
    const int nyh = p.ny >> 1;
    // FS: rows 0 .. ny/2 - 1 fill whole units; the Nyquist rows (k1 = ny/2) of GX consecutive slabs share one extra unit (transform A
    // of group g = slab s0 + g; B idles) instead of a fifth unit per slab with one live row in 32 (dft (1024, 65536): +12 %)
    const int upr = FS ? nyh / RPU : p.nrow_pad / RPU;  // units per slab
    const bool nyq = FS && (int)blockIdx.x >= p.nslab * upr;
    const int s0 = nyq ? ((int)blockIdx.x - p.nslab * upr) * GX : 0;
    const int slab = nyq ? min(s0 + g, p.nslab - 1) : (int)blockIdx.x / upr, unit = nyq ? 0 : (int)blockIdx.x % upr, ky0 = nyq ? nyh : unit * RPU;
    // rows beyond ny/2 (padding of the last unit) are computed on row ny/2's data and never stored or binned
    const int kyA = min(ky0 + g, nyh), kyB = TWO ? kyA : min(ky0 + GX + g, nyh);
    // uniform 64-bit bases + 32-bit per-lane byte offsets (scalar-base loads).  The residual-trend pairs go first: 16 loads
    // in flight beside the 32 of the rows (issued after them they came in four serialised batches: +4.5 us per slab)
    const char* __restrict__ w2s = reinterpret_cast<const char*>(p.w2 + (size_t)slab * p.nrow_pad * NX);
    const char* __restrict__ w2t = TWO ? reinterpret_cast<const char*>(p.w2b + (size_t)slab * p.nrow_pad * NX) : w2s;
    const char* __restrict__ crb = reinterpret_cast<const char*>(p.corr + (size_t)slab * NX * 2);
    const bool addback = p.detrend && !(XRFT_YDBG & 1);
    cf cr[16];
    if (addback) {
#pragma unroll
        for (int q = 0; q < 16; ++q) cr[q] = *reinterpret_cast<const cf*>(crb + (unsigned)(u + NT * q) * 8u);
    }
    const unsigned offA = w2_offset(p, kyA, u) * 8u, offB = w2_offset(p, kyB, u) * 8u;
    cf a[16], b[16];
    const bool ntw = (XRFT_YTUNE(p) & 8) != 0;
    if (NT >= (1 << p.l_cw)) {  // x = u + NT q advances by whole column blocks: constant stride
        const unsigned qstr = (unsigned)(((NT >> p.l_cw) * 2) << (p.l_rk + p.l_2gy)) * 8u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = xrft_load8_pol(w2s + (offA + qstr * (unsigned)q), ntw);
            b[q] = xrft_load8_pol(w2t + (offB + qstr * (unsigned)q), ntw);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const cf*>(w2s + w2_offset(p, kyA, u + NT * q) * 8u);
            b[q] = *reinterpret_cast<const cf*>(w2t + w2_offset(p, kyB, u + NT * q) * 8u);
        }
    }
    if (addback && W2D) {
        // four-step with a window: column x of the [ny][nx] view has its own window w[nx i1 + x], so the transforms of the window
        // (and of the window times i1 - ibar) that carry the residual line back are tables over (x, k1): [x][k1 < nrow_pad]
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float al = cr[q].re, ga = cr[q].im;
            const cf* w0 = p.what0 + (size_t)(u + NT * q) * p.nrow_pad;
            const cf* w1 = p.what1 + (size_t)(u + NT * q) * p.nrow_pad;
            const cf a0 = w0[kyA], a1 = w1[kyA], b0 = w0[kyB], b1 = w1[kyB];
            a[q].re = fmaf(al, a0.re, fmaf(ga, a1.re, a[q].re));
            a[q].im = fmaf(al, a0.im, fmaf(ga, a1.im, a[q].im));
            b[q].re = fmaf(al, b0.re, fmaf(ga, b1.re, b[q].re));
            b[q].im = fmaf(al, b0.im, fmaf(ga, b1.im, b[q].im));
        }
    } else if (addback) {  // add back wx[x] * (subtracted line - plane fit) in the spectral domain (see fasty_cols_kernel)
        const cf a0 = p.what0[kyA], a1 = p.what1[kyA], b0 = p.what0[kyB], b1 = p.what1[kyB];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float al = cr[q].re, ga = cr[q].im;
            a[q].re = fmaf(al, a0.re, fmaf(ga, a1.re, a[q].re));
            a[q].im = fmaf(al, a0.im, fmaf(ga, a1.im, a[q].im));
            if (!TWO) {
                b[q].re = fmaf(al, b0.re, fmaf(ga, b1.re, b[q].re));
                b[q].im = fmaf(al, b0.im, fmaf(ga, b1.im, b[q].im));
            }
        }
        if (TWO) {  // the second field has its own residual trend (these loads are not hidden: a cross spectrum with detrending pays for them)
            const char* __restrict__ crc = reinterpret_cast<const char*>(p.corr_b + (size_t)slab * NX * 2);
#pragma unroll
            for (int q = 0; q < 16; ++q) cr[q] = *reinterpret_cast<const cf*>(crc + (unsigned)(u + NT * q) * 8u);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                b[q].re = fmaf(cr[q].re, b0.re, fmaf(cr[q].im, b1.re, b[q].re));
                b[q].im = fmaf(cr[q].re, b0.im, fmaf(cr[q].im, b1.im, b[q].im));
            }
        }
    }
    if (FS) {  // x W_N^(i2 k1), i2 = u + NT q: W^(k1 u) (W^(k1 NT))^q -- two table loads and a product tree per row
        const cf ba = p.tw_big[kyA * u], sa = p.tw_big[kyA * NT], bb_ = p.tw_big[kyB * u], sb = p.tw_big[kyB * NT];
        twiddle16(a, sa);
        twiddle16(b, sb);
#pragma unroll
        for (int q = 0; q < 16; ++q) { a[q] = cmul(a[q], ba); b[q] = cmul(b[q], bb_); }
    }
    if (!(XRFT_YDBG & 4)) fft_p2_pair<NX>(a, b, u, mine, p.tw_x, tw2);
    if (TWO) {  // F0 conj(F1) * scale, in place of transform A (xrft.py:825)
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = cscale(cmulc(a[e], b[e]), p.scale);
    }
    const int mx = NX - 1, my = p.ny - 1, sx = p.shift_x;
    // ---- store loops.  Power rows staged as floats at sbase[rl * RSP + nat16(kx)] (rl < nrows) are the unit's rows r0 .. r0 + nrows - 1;
    // every valid row leaves twice: rotated (direct) and reversed + rotated (mirror); 16-byte stores, whole rows
    constexpr int RSP = R::RS + (FS ? 1 : 0);
    auto store_power = [&](const float* sbase, int r0, int nrows) {
        if (p.half) {  // rows of nx/2 + 1 samples (an odd length: 4-byte stores, still whole lines per wave); row -ky reads the row backwards
            constexpr int W = NX / 2 + 1;
            float* __restrict__ oh = reinterpret_cast<float*>(p.out) + (size_t)slab * p.ny * W;
            for (int e = tid; e < nrows * 2 * W; e += THR) {
                const int kx = e % W, rr = e / W, rl = rr >> 1, mir = rr & 1;
                const int ky = ky0 + r0 + rl;
                if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
                float v = sbase[rl * RSP + nat16(mir ? (NX - kx) & mx : kx)];
                if (p.realdim2 && kx != 0 && kx != NX / 2) v *= 2.0f;
                oh[(size_t)(mir ? p.ny - ky : ky) * W + kx] = v;
            }
            return;
        }
        float* __restrict__ outs = reinterpret_cast<float*>(p.out) + (size_t)slab * p.ny * NX;
        constexpr int CPR = NX / 4;  // float4 chunks per row
        for (int e = tid; e < nrows * 2 * CPR; e += THR) {
            const int chunk = e % CPR, rr = e / CPR, rl = rr >> 1, mir = rr & 1;
            const int ky = ky0 + r0 + rl;
            if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
            const float* row = sbase + rl * RSP;
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
            // non-temporal: the result is not read again, and keeping it out of the caches leaves the Infinity Cache to the
            // intermediate (scripts/ubench/yfirst.hip: 45.8 vs 50.9 us per slab for the two passes at 2 slabs per group)
            if (!(XRFT_YDBG & 8) || v.x == 1.2345f) xrft_store_pol(reinterpret_cast<char*>(outs), (unsigned)(((size_t)orow * NX + c) * 4u), v, (XRFT_YTUNE(p) >> 4) & 1 ? 1 : 0);
        }
    };
    // complex rows staged at cbase[rl * RSC + nat16(kx)] (natural order) = the unit's rows r0 .. r0 + nrows - 1 (not the four-step form)
    constexpr int RSC = NX + NX / 16 + (FS ? 1 : 0);
    auto store_complex = [&](const cf* cbase, int r0, int nrows) {
        if (p.half) {  // kx = 0..nx/2 only, unshifted; F(-ky, kx) = conj F(ky, -kx)
            constexpr int W = NX / 2 + 1;
            for (int e = tid; e < nrows * 2 * W; e += THR) {
                const int kx = e % W, rr = e / W, rl = rr >> 1, mir = rr & 1;
                const int ky = ky0 + r0 + rl;
                if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
                cf v = cbase[rl * RSC + nat16(mir ? (NX - kx) & mx : kx)];
                if (mir) v = cconj(v);
                const int fy = mir ? p.ny - ky : ky;
                if (p.ph_on) v = cmul(v, cmul(p.ph_y[fy], p.ph_x[kx]));
                if (p.realdim2 && kx != 0 && kx != NX / 2) v = cscale(v, 2.0f);
                const size_t o = ((size_t)slab * p.ny + fy) * W + kx;
                if (MODE == 3) reinterpret_cast<float*>(p.out)[o] = (float)atan2((double)v.im, (double)v.re);
                else reinterpret_cast<cf*>(p.out)[o] = v;
            }
            return;
        }
        typedef typename std::conditional<MODE == 3, float, cf>::type OutT;
        OutT* __restrict__ outs = reinterpret_cast<OutT*>(p.out) + (size_t)slab * p.ny * NX;
        constexpr int CPR = NX / 2;  // pairs of samples per row
        for (int e = tid; e < nrows * 2 * CPR; e += THR) {
            const int chunk = e % CPR, rr = e / CPR, rl = rr >> 1, mir = rr & 1;
            const int ky = ky0 + r0 + rl;
            if (ky > nyh || (mir && (ky == 0 || ky == nyh))) continue;
            const cf* row = cbase + rl * RSC;
            const int c = 2 * chunk;
            const int fx0 = (c - sx) & mx, fx1 = (c + 1 - sx) & mx;  // unshifted frequency indices of the two output columns
            const int fy = mir ? (p.ny - ky) & my : ky;
            cf v0, v1;
            if (!mir) { v0 = row[nat16(fx0)]; v1 = row[nat16(fx1)]; }
            else { v0 = cconj(row[nat16((NX - fx0) & mx)]); v1 = cconj(row[nat16((NX - fx1) & mx)]); }  // F(-k) = conj F(k)
            if (p.ph_on) {
                const cf py = p.ph_y[fy];
                v0 = cmul(v0, cmul(py, p.ph_x[fx0]));
                v1 = cmul(v1, cmul(py, p.ph_x[fx1]));
            }
            const size_t o = (size_t)((fy + p.shift_y) & my) * NX + c;
            if (MODE == 3) {  // cross phase (xrft.py:838-874)
                struct alignas(8) P2f { float x, y; } ang;
                ang.x = (float)atan2((double)v0.im, (double)v0.re); ang.y = (float)atan2((double)v1.im, (double)v1.re);
                *reinterpret_cast<P2f*>(reinterpret_cast<float*>(outs) + o) = ang;
            } else {
                xrft_store_nt2(reinterpret_cast<cf*>(outs) + o, v0, v1);
            }
        }
    };
    if constexpr (ISO) {
        if (p.tfirst != nullptr) {
            // Radial sums (xrft.py:895-906) of a RADIAL bin map -- along a row the bin depends on |kx| only and never decreases
            // with it, and the Hermitian twin (-ky, -kx) of a sample falls into the sample's bin (verified on the host,
            // fasty_build_tcodes) -- with no atomic at all: the bins of a row are contiguous ranges of kx on either side of
            // kx = 0.  (1) Every 16-sample segment of the staged rows is reduced in place: the sum of each run of equal bins
            // lands on the run's last sample (float32 adds in sample order; the run ends come from a 16-bit step mask).
            // (2) The owner of a bin walks its two kx ranges in every row, one staged value per segment it touches, and adds them
            // in float64 in a fixed order: bit-reproducible, nothing to zero, no per-bin exponent pass, inf / nan propagate as in
            // any floating-point sum.  (Atomics -- int64 fixed point, the only way to make them order-independent -- cost 10 of
            // the row pass's 30 us per 4096^2 slab even issued once per run: profiles/r03_tune_iso.txt.)
            constexpr int NRW = MODE == 1 ? 2 * GX : GX;      // staged rows
            constexpr int RSI = MODE == 1 ? RSP : 2 * RSC;    // floats per staged row
            constexpr int CPS = MODE == 1 ? 1 : 2;            // floats per sample
            constexpr int SPR = NX / 16;                      // segments per row
            if (MODE == 1) {
#pragma unroll
                for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                    for (int k3 = 0; k3 < G::R3; ++k3) {
                        const int sl = nat16(held_k<NX>(u, bb, k3));
                        const cf va = a[bb * G::R3 + k3], vb = b[bb * G::R3 + k3];
                        stg[g * RSP + sl] = (va.re * va.re + va.im * va.im) * p.scale;
                        stg[(GX + g) * RSP + sl] = (vb.re * vb.re + vb.im * vb.im) * p.scale;
                    }
            } else {
#pragma unroll
                for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                    for (int k3 = 0; k3 < G::R3; ++k3) lds[g * RSC + nat16(held_k<NX>(u, bb, k3))] = a[bb * G::R3 + k3];
            }
            __syncthreads();
            if (p.out != nullptr) {  // the spectrum leaves first: step (1) overwrites the staged samples
                if (MODE == 1) store_power(stg, 0, NRW); else store_complex(lds, 0, NRW);
                __syncthreads();
            }
            for (int sg = tid; sg < NRW * SPR; sg += THR) {
                const int row = sg / SPR, s16 = sg % SPR, ky = ky0 + row;
                if (ky > nyh) continue;
                const unsigned mask = p.tcodes[(size_t)ky * SPR + s16] >> 16;  // bit i: the bin changes between samples i - 1 and i
                float* q = stg + row * RSI + CPS * (17 * s16);                  // nat16(16 s16) = 17 s16: the segment is contiguous
                // all 16 samples first, the running sums in registers, every position written back (only the run ends are read again):
                // no branch and ONE trip to the LDS -- a read, a conditional write and a wait per sample was a chain of 16 LDS latencies
                // per segment, 128 per thread, most of what the radial sums added to the row pass
                float vr[16], vi[MODE == 2 ? 16 : 1];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    vr[i] = q[CPS * i];
                    if (MODE == 2) vi[MODE == 2 ? i : 0] = q[2 * i + 1];
                }
                float sr = 0.f, si = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sr += vr[i];
                    vr[i] = sr;
                    if (MODE == 2) { si += vi[MODE == 2 ? i : 0]; vi[MODE == 2 ? i : 0] = si; }
                    const bool end = ((mask >> (i + 1)) & 1u) != 0u;
                    sr = end ? 0.f : sr;
                    si = end ? 0.f : si;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    q[CPS * i] = vr[i];
                    if (MODE == 2) q[2 * i + 1] = vi[MODE == 2 ? i : 0];
                }
            }
            __syncthreads();
            double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * p.nbins * HW;
            // task = (bin, row): NRW adjacent lanes share a bin, one row each, and their sums meet in lane order (a bin per thread leaves the
            // one thread whose bin hugs |k| = ky -- ranges of up to 300 samples in every row -- working alone); radial_gather_batch above
            static_assert((NRW & (NRW - 1)) == 0 && NRW <= 16 && THR % NRW == 0, "rows per workgroup");
            constexpr int BPI = THR / NRW, KB = MODE == 2 ? 2 : 3;
            const int row = tid % NRW, ky = ky0 + row, blane = tid / NRW;
            const bool live = ky <= nyh, twin = ky != 0 && ky != nyh;
            const float* rowp = stg + row * RSI;
            // only the bins the unit's rows reach, |k| = ky0 dky .. |(ky0 + NRW - 1, nx/2)| (p.twin, from the map itself): 46 % of them
            // on average -- the range table and the partial sums of all bins for every 4 rows of a 4096^2 slab were 25 MB of traffic
            // beside the 67 MB of the rows
            const unsigned bw = p.twin[unit];
            const int blo = (int)(bw & 0xffffu), bhi = (int)(bw >> 16);
            const unsigned short* __restrict__ frow = p.tfirst + (size_t)min(ky, nyh) * (p.nbins + 1);
            auto range_of = [&](int bn_) -> unsigned {  // the bin holds |kx| = s .. e - 1 of this row: s | e << 16
                if (!(live && bn_ < bhi)) return 0u;
#ifdef XRFT_EMULATE
                return (unsigned)frow[bn_] | ((unsigned)frow[bn_ + 1] << 16);
#else
                return reinterpret_cast<const U16Pair*>(frow + bn_)->v;
#endif
            };
            const int nsw = (bhi - blo + BPI - 1) / BPI;  // sweeps of this unit (uniform)
            unsigned rg[KB], rn[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) rn[k] = range_of(blo + k * BPI + blane);
            for (int k0 = 0; k0 < nsw; k0 += KB) {
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    rg[k] = rn[k];
                    rn[k] = range_of(blo + (k0 + KB + k) * BPI + blane);  // (the next batch's ranges are in flight behind this batch's sums)
                }
                radial_gather_batch<NX, NRW, CPS, BPI, KB>(rowp, row, live, twin, blo, bhi, k0, min(nsw - k0, KB), rg, blane, part, true);
            }
            return;
        }
        // Radial sums (xrft.py:895-906), bit-reproducible: floating-point atomics would make a sum depend on the order in which waves
        // arrive.  The results are staged in LDS in natural order, half of the workgroup's rows per round (the other half of the
        // transforms' LDS holds the tables), and a thread owns L CONSECUTIVE kx of one row: neighbouring samples mostly share a
        // radial bin, so the thread reduces RUNS of equal (bin, mirror bin) pairs in registers (float32 adds in sample order: as
        // reproducible as the transform itself) and touches the tables once per run, and the lanes of a wave (L samples apart)
        // mostly hit different bins.  (In register order -- a thread's samples 16 apart, its
        // neighbours' 1 apart -- every sample was its own atomic and a wave collided in a handful of bins: 33 us per 4096^2 slab
        // with nothing stored, profiles/r02_bench_configs.txt.)  Two sweeps per round: (1) the largest run sum per bin (atomicMax
        // on the float bits: order-independent), (2) every run sum converted to int64 fixed point 40 bits below its bin's maximum
        // and added with INTEGER atomics (exact, order-independent).  A value at (ky, kx) goes to its bin and once more (conjugated) to the
        // bin of (-ky, -kx).  The workgroup's per-bin sums go to a partial table that iso_reduce_kernel adds in unit order.
        constexpr int NR = MODE == 1 ? GX : GX / 2;    // rows per round
        constexpr int L = NR * NX / THR;                // 16 (power), 8 (cross) consecutive samples per thread
        constexpr int RSI = MODE == 1 ? RSP : 2 * RSC;  // floats per staged row
        static_assert(GX >= 2 && L * THR == NR * NX && (16 % L) == 0, "radial-sum geometry");
        double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * p.nbins * HW;
        const int rl = tid / (NX / L), kx0 = (tid % (NX / L)) * L;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {  // (unrolled: transform A's registers are dead in round 2)
            float* sreg = stg + (rd ? NR * RSI : 0);
            unsigned long long* acc = reinterpret_cast<unsigned long long*>(stg + (rd ? 0 : NR * RSI));  // [nbins][HW] int64 fixed-point sums
            unsigned* bmax = reinterpret_cast<unsigned*>(acc + p.nbins * HW);                              // [nbins] float bits of the largest magnitude
            if (MODE == 1) {
#pragma unroll
                for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                    for (int k3 = 0; k3 < G::R3; ++k3) {
                        const cf v = rd ? b[bb * G::R3 + k3] : a[bb * G::R3 + k3];
                        sreg[g * RSP + nat16(held_k<NX>(u, bb, k3))] = (v.re * v.re + v.im * v.im) * p.scale;
                    }
            } else if (g / NR == rd) {
                cf* creg = reinterpret_cast<cf*>(sreg);
#pragma unroll
                for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                    for (int k3 = 0; k3 < G::R3; ++k3) creg[(g % NR) * RSC + nat16(held_k<NX>(u, bb, k3))] = a[bb * G::R3 + k3];
            }
            for (int i = tid; i < p.nbins * HW; i += THR) acc[i] = 0ull;
            for (int i = tid; i < p.nbins; i += THR) bmax[i] = 0u;
            // this thread's run of L samples of row ky0 + rd NR + rl: bin codes in natural order, (bin + 1) | (mirror bin + 1) << 16
            unsigned codes[L];
            const int kyr = ky0 + rd * NR + rl;
            if (XRFT_YTUNE(p) & (1 << 16)) {  // (ablation, tuning build: no bin-code loads)
#pragma unroll
                for (int i = 0; i < L; ++i) codes[i] = (unsigned)(1 + ((kx0 + i) >> 2)) * 0x10001u;
            } else if (p.tcodes_compact) {
                // A radial bin map is monotone along a half row and moves by at most one bin per sample: 16 samples = the first
                // one's bin and a mask of the steps (built and verified on the host, fasty_build_tcodes).  4 bytes per 16 samples
                // instead of 64: the full table (33.6 MB at 4096^2, read once per slab -- loads do not stay in the Infinity Cache --
                // cost 7.3 of the row pass's 29.6 us, profiles/r03_tune_iso.txt) becomes 2 MB that live in L2.  The mirror sample
                // (-ky, -kx) is in the same bin (verified too) except on the rows ky = 0 and ny/2, which are their own mirrors.
                const unsigned w = p.tcodes[(size_t)kyr * (NX / 16) + (kx0 >> 4)];
                const unsigned first = w & 0xffffu, mask = w >> 16;
                const bool up = kx0 < NX / 2, twin = kyr != 0 && kyr != nyh;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    const int pos = (kx0 & 15) + i;  // position inside the 16-sample segment
                    const unsigned steps = (unsigned)__popc(mask & ((2u << pos) - 1u));
                    const unsigned cd = first ? (up ? first + steps : first - steps) : 0u;
                    codes[i] = cd | (twin ? cd << 16 : 0u);
                }
            } else {
                const unsigned* __restrict__ tc = p.tcodes + (size_t)kyr * NX + kx0;
#pragma unroll
                for (int i = 0; i < L; i += 4) {
                    const uint4 c4 = *reinterpret_cast<const uint4*>(tc + i);
                    codes[i] = c4.x; codes[i + 1] = c4.y; codes[i + 2] = c4.z; codes[i + 3] = c4.w;
                }
            }
            __syncthreads();
            // sweep 1: the thread adds up each run of equal codes (float32, in sample order: a fixed order), keeps the sum at the
            // run's last sample, and records its magnitude in the bins it goes to (one atomicMax per run and bin)
            const float* src = sreg + rl * RSI + (MODE == 1 ? 1 : 2) * nat16(kx0);
            float rsr[L], rsi[MODE == 2 ? L : 1];
            if (!(XRFT_YTUNE(p) & (1 << 17))) {
                float sr = 0.f, si = 0.f;
#pragma unroll
                for (int i = 0; i < L; ++i) {
                    if (MODE == 1) sr += src[i];
                    else { sr += src[2 * i]; si += src[2 * i + 1]; }
                    rsr[i] = sr;
                    if (MODE == 2) rsi[i] = si;
                    const bool last = i == L - 1 || codes[i + 1] != codes[i];
                    if (last) {
                        const unsigned cd = codes[i] & 0xffffu, cm = codes[i] >> 16, m = __float_as_uint(fabsf(sr) + fabsf(si));
                        if (cd) atomicMax(&bmax[cd - 1], m);
                        if (cm && cm != cd) atomicMax(&bmax[cm - 1], m);
                        sr = 0.f; si = 0.f;
                    }
                }
            }
            __syncthreads();
            // sweep 2: every run sum converted to int64 fixed point 40 bits below its bin's largest and added with an INTEGER atomic
            // (exact: the order in which lanes arrive does not matter)
            if (!(XRFT_YTUNE(p) & (1 << 17)))
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const bool last = i == L - 1 || codes[i + 1] != codes[i];
                if (last) {
                    const unsigned cd = codes[i] & 0xffffu, cm = codes[i] >> 16;
                    const float re = rsr[i], im = MODE == 2 ? rsi[MODE == 2 ? i : 0] : 0.f;
                    if (cd) {
                        const int eb = (int)(bmax[cd - 1] >> 23);
                        atomicAdd(&acc[HW * (cd - 1)], (unsigned long long)(fixed40(re, eb) * (cd == cm ? 2 : 1)));
                        if (MODE == 2 && cd != cm) atomicAdd(&acc[2 * (cd - 1) + 1], (unsigned long long)fixed40(im, eb));  // (cd == cm: V + conj V is real)
                    }
                    if (cm && cm != cd) {
                        const int eb = (int)(bmax[cm - 1] >> 23);
                        atomicAdd(&acc[HW * (cm - 1)], (unsigned long long)fixed40(re, eb));
                        if (MODE == 2) atomicAdd(&acc[2 * (cm - 1) + 1], (unsigned long long)fixed40(-im, eb));
                    }
                }
            }
            __syncthreads();
            // this round's sums, back in floating point, into the workgroup's row of the partial table (the same thread adds round 2
            // to what it wrote in round 1: a fixed order).  A bin with an inf / nan member is +inf (power) or nan, as IEEE sums are.
            if (!(XRFT_YTUNE(p) & (1 << 18)))
            for (int i = tid; i < p.nbins * HW; i += THR) {
                const unsigned bm = bmax[i / HW];
                double v = ldexp((double)(long long)acc[i], (int)(bm >> 23) - 127 - 40);
                if ((bm >> 23) == 0xffu) v = __longlong_as_double((MODE == 1 && bm == 0x7f800000u) ? 0x7ff0000000000000ll : 0x7ff8000000000000ll);
                part[i] = rd ? part[i] + v : v;
            }
            if (p.out != nullptr) {
                if (MODE == 1) store_power(sreg, rd * NR, NR);
                else store_complex(reinterpret_cast<const cf*>(sreg), rd * NR, NR);
            }
            __syncthreads();  // the next round's tables overwrite this round's staging
        }
        return;
    }
    if (MODE == 1 && p.out != nullptr) {
        // power, staged row-major [row][kx] in natural order with the conflict-free 17/16 padding (FS: + 1, the transposed
        // read-out runs down the rows)
#pragma unroll
        for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
            for (int k3 = 0; k3 < G::R3; ++k3) {
                const int s = nat16(held_k<NX>(u, bb, k3));
                const cf va = a[bb * G::R3 + k3], vb = b[bb * G::R3 + k3];
                stg[g * RSP + s] = (va.re * va.re + va.im * va.im) * p.scale;
                stg[(GX + g) * RSP + s] = (vb.re * vb.re + vb.im * vb.im) * p.scale;
            }
        __syncthreads();
        if (FS) {  // sample k = k1 + ny k2 (fftshift: k2 + nx/2): the unit's RPU rows are consecutive samples
            float* __restrict__ ob = reinterpret_cast<float*>(p.out);
            for (int e = tid; e < RPU * NX * 2; e += THR) {
                const int r = e % RPU, rest = e / RPU, mir = rest & 1, k2 = rest >> 1;
                const int k1 = nyq ? nyh : ky0 + r;
                if (nyq ? (r >= GX || s0 + r >= p.nslab || mir) : (mir && k1 == 0)) continue;  // (Nyquist unit: row r = slab s0 + r)
                const float v = stg[r * RSP + nat16(k2)];
                const int o1k = mir ? p.ny - k1 : k1, o2k = mir ? (NX - 1 - k2) : k2;  // X[N - k]: (ny - k1) + ny (nx - 1 - k2)
                ob[(size_t)(nyq ? s0 + r : slab) * p.ny * NX + (size_t)((o2k + sx) & mx) * p.ny + o1k] = v;
            }
        } else {
            store_power(stg, 0, RPU);
        }
    }
    if (MODE != 1 && p.out != nullptr) {
        // complex results: GX rows at a time staged in natural order (a round fills the transforms' LDS exactly); the complex
        // spectrum of one field takes two rounds (transform A's rows, then B's)
        cf* cstg = lds;
        constexpr int NROUND = TWO ? 1 : 2;
#pragma unroll
        for (int round = 0; round < NROUND; ++round) {
            if (round) __syncthreads();
            const float sc = TWO ? 1.0f : p.scale;
#pragma unroll
            for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                for (int k3 = 0; k3 < G::R3; ++k3)
                    cstg[g * RSC + nat16(held_k<NX>(u, bb, k3))] = cscale(round ? b[bb * G::R3 + k3] : a[bb * G::R3 + k3], sc);
            __syncthreads();
            if (FS) {  // transposed: sample k = k1 + ny k2; the true-phase table is indexed by the unshifted sample index
                cf* __restrict__ ob = reinterpret_cast<cf*>(p.out);
                for (int e = tid; e < GX * NX * 2; e += THR) {
                    const int r = e % GX, rest = e / GX, mir = rest & 1, k2 = rest >> 1;
                    const int k1 = nyq ? nyh : ky0 + round * GX + r;
                    if (nyq ? (round != 0 || s0 + r >= p.nslab || mir) : (mir && k1 == 0)) continue;  // (Nyquist unit: row r = slab s0 + r, transform A only)
                    cf v = cstg[r * RSC + nat16(k2)];
                    if (mir) v = cconj(v);
                    const int o1k = mir ? p.ny - k1 : k1, o2k = mir ? (NX - 1 - k2) : k2;
                    if (p.ph_on) v = cmul(v, p.ph_x[o2k * p.ny + o1k]);
                    ob[(size_t)(nyq ? s0 + r : slab) * p.ny * NX + (size_t)((o2k + sx) & mx) * p.ny + o1k] = v;
                }
                continue;
            }
            store_complex(cstg, round * GX, GX);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// plane fit from the per-column sums (one 256-thread block per slab, float64, fixed summation order): column means
// m_x = sum d / ny and slopes s_x = sum (i - ibar) d / sum (i - ibar)^2; a = mean(m_x), b = slope of m_x over x, c = mean(s_x)
// (the centred regressors of a full grid are orthogonal, so this IS the least-squares plane of xrft/detrend.py:100-113).
// Output: corr[x] = wx[x] * (line pass 1 subtracted - plane) as (offset at ibar, slope).
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) fasty_fit_kernel(const double* colfit, const float* win_x, float* corr, int nx, int ny, int detrend) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x, tid = threadIdx.x;
    const double* cf4 = colfit + (size_t)slab * nx * 4;
    const double xbar = 0.5 * (nx - 1), sxx = (double)nx * ((double)nx * nx - 1.0) / 12.0;
    const double inv_n = 1.0 / ny, inv_sii = 12.0 / ((double)ny * ((double)ny * ny - 1.0));
    double s[3] = {0.0, 0.0, 0.0};
    for (int x = tid; x < nx; x += 256) {
        const double m = cf4[4 * x] * inv_n + cf4[4 * x + 2], sl = cf4[4 * x + 1] * inv_sii + cf4[4 * x + 3];  // (residual sums + the subtracted line)
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
    float* out = corr + (size_t)slab * nx * 2;
    for (int x = tid; x < nx; x += 256) {
        const double wx = win_x[x];
        out[2 * x] = (float)(wx * (cf4[4 * x + 2] - a - b * ((double)x - xbar)));
        out[2 * x + 1] = (float)(wx * (cf4[4 * x + 3] - c));
    }
}

// ------------------------------------------------------------------------------------------------
// four-step 1-D: the trend of the WHOLE sequence (mean, or the least-squares line a + b (n - nbar), scipy.signal.detrend,
// xrft/detrend.py:64-71) from the per-column sums of its [ny][nx] view, n = nx i1 + i2:
//   sum d = sum_x S0[x],   sum (n - nbar) d = sum_x ( nx (S1[x] + ibar S0[x]) + x S0[x] ) - nbar sum d
// Along column x the line is [a + b (x - nbar + nx ibar)] + (b nx) (i1 - ibar): corr[x] = subtracted line - that (no window).
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) fasty_fit1d_kernel(const double* colfit, float* corr, int nx, int ny, int detrend) {
    XRFT_DYN_SMEM(smem_raw);
    double* red = reinterpret_cast<double*>(smem_raw);
    const int slab = blockIdx.x, tid = threadIdx.x;
    const double* cf4 = colfit + (size_t)slab * nx * 4;
    const double N = (double)nx * (double)ny, nbar = 0.5 * (N - 1.0), ibar = 0.5 * (ny - 1), sii = (double)ny * ((double)ny * ny - 1.0) / 12.0;
    double s[3] = {0.0, 0.0, 0.0};
    for (int x = tid; x < nx; x += 256) {
        const double s0 = cf4[4 * x] + (double)ny * cf4[4 * x + 2], s1 = cf4[4 * x + 1] + cf4[4 * x + 3] * sii;  // (residual sums + the subtracted line)
        s[0] += s0;
        s[1] += (double)nx * (s1 + ibar * s0) + (double)x * s0;
    }
    block_sum<3>(s, red);
    __syncthreads();
    if (tid == 0) { red[0] = s[0]; red[1] = s[1]; }
    __syncthreads();
    const double a = red[0] / N;
    const double b = detrend == 2 ? (red[1] - nbar * red[0]) / (N * (N * N - 1.0) / 12.0) : 0.0;
    float* out = corr + (size_t)slab * nx * 2;
    for (int x = tid; x < nx; x += 256) {
        out[2 * x] = (float)(cf4[4 * x + 2] - (a + b * ((double)x - nbar + (double)nx * ibar)));
        out[2 * x + 1] = (float)(cf4[4 * x + 3] - b * (double)nx);
    }
}

}  // namespace xrft

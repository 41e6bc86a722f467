// fasty_iso.h -- pass 2 of the y-first pipeline (fasty.h) when NOTHING but the radial sums leaves it:
//     isotropic_power_spectrum / isotropic_cross_spectrum (xrft/xrft.py:1013-1187 -> isotropize :948-1010 -> _groupby_bins_agg :910-945)
// on a RADIAL bin map (the bins of a half row are contiguous ranges of |kx|: fasty_build_tcodes verifies it on the host).
//
// fasty_rows_kernel<NX, MODE, ISO> with p.out == nullptr was 21.2 us per 4096^2 slab against 10.7 us for reading the 67-MB half
// spectrum: with no result to store the kernel is a chain load -> transform -> radial sums per workgroup, two workgroups per CU,
// and nothing of a unit's own overlaps (VALU busy 45 %, memory busy 60 %).  This kernel is the same arithmetic in a different order:
//   * PERSISTENT workgroups (as many as fit the chip, the host asks the occupancy calculator): the stage-2 twiddle table and the
//     lane's first-stage twiddle are loaded once, not once per four rows (each was an L2 round trip in front of a unit's loads);
//   * the NEXT unit's rows are requested as soon as the transforms' registers are free -- behind the |F|^2 staging, in front of
//     the radial sums -- so a workgroup's loads fly while it sums and while the other residents of the CU transform;
//   * everything the radial sums read from global memory (segment masks, the unit's bin window, the bins' |kx| ranges) is
//     requested BEFORE those rows: loads return in order, a table load behind the prefetch would wait for all of it.
// The sums themselves are fasty_rows_kernel's per-bin gather (no atomics, fixed order, bit-reproducible): see the comments there.
#pragma once
#include "fasty.h"

namespace xrft {

#ifndef XRFT_EMULATE
#define XRFT_ISO_CLOCK() ((long long)__builtin_readcyclecounter())
#else
#define XRFT_ISO_CLOCK() (0ll)
#endif

// MODE 1: power (two rows of one field per thread), MODE 2: cross (the same row of the two fields; F0 conj(F1) in registers).
// TIM: the profiling build of the kernel -- wave 0 of every workgroup adds up the shader-clock cycles of its phases into
// p.tim[block][8] (scripts/prof.py iso-phases); never launched by the product path.
template <int NX, int MODE, bool TIM = false>
__global__ void __launch_bounds__((YRows<NX, false>::THR), 4) fasty_isorows_kernel(FastY p) {
    static_assert(MODE == 1 || MODE == 2, "radial sums exist for power and cross spectra");
    static_assert(NX >= 1024, "rows of 1024 samples and more (NX / 16 threads per row span whole column blocks of pass 1; shorter rows keep fasty_rows_kernel)");
    typedef P2<NX> G;
    typedef YRows<NX, false> R;
    constexpr bool TWO = MODE == 2;
    constexpr int NT = G::NT, GX = R::GX, THR = R::THR, NRW = TWO ? GX : 2 * GX, GSTR = YLds<NX, GX>::GSTR;
    constexpr int HW = TWO ? 2 : 1, CPS = TWO ? 2 : 1, SPR = NX / 16;
    constexpr int RSP = R::RS, RSC = NX + NX / 16, RSI = TWO ? 2 * RSC : RSP;  // floats per staged row
    constexpr int NSEG = NRW * SPR / THR;  // 16-sample segments per thread: 2 (power), 1 (cross)
    constexpr int BPI = THR / NRW;         // bins per sweep of the gather (NRW adjacent lanes share a bin, one row each)
    constexpr int KB = TWO ? 2 : 3;        // sweeps of the gather in flight together (their staged values live beside the prefetched rows: 4 CPS floats each)
    constexpr int KPRE = 6;                // sweeps whose |kx| ranges are requested in front of the prefetch (nfactor = 4: at most 6)
    static_assert(NSEG * THR == NRW * SPR && NSEG >= 1 && (NRW & (NRW - 1)) == 0 && NRW <= 64 && THR % NRW == 0, "radial-sum geometry");
    static_assert((size_t)NRW * RSI * sizeof(float) <= (size_t)GX * GSTR * sizeof(cf), "the staged rows alias the transforms' LDS and leave the twiddle table alone");
    XRFT_DYN_SMEM(smem_raw);
    cf* lds = reinterpret_cast<cf*>(smem_raw);
    float* stg = reinterpret_cast<float*>(smem_raw);
    cf* tw2 = lds + GX * GSTR;
    const cf w1 = p.tw_x[threadIdx.x / GX];  // W_NX^u: once per workgroup
    fill_tw2<NX>(tw2, p.tw_x, threadIdx.x, THR);
    const int nyh = p.ny >> 1, upr = p.nrow_pad / NRW, total = p.nslab * upr;
    const bool addback = p.detrend != 0;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    auto lap = [&](int i, bool landed) {
        if (TIM) {
#ifndef XRFT_EMULATE
            if (landed) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the phase ends when its loads have landed
#endif
            const long long t = XRFT_ISO_CLOCK();
            tacc[i] += t - tlast;
            tlast = t;
        }
    };

    cf a[16], b[16];
    // the rows of unit `un` (= slab * upr + unit of NRW rows): uniform 64-bit bases + 32-bit per-lane byte offsets (scalar-base loads)
    auto load_rows = [&](int un, int g, int u) {
        const int slab = un / upr, ky0 = (un % upr) * NRW;
        // rows beyond ny/2 (padding of the last unit) are computed on row ny/2's data and never binned
        const int kyA = min(ky0 + g, nyh), kyB = TWO ? kyA : min(ky0 + GX + g, nyh);
        const char* __restrict__ w2s = reinterpret_cast<const char*>(p.w2 + (size_t)slab * p.nrow_pad * NX);
        const char* __restrict__ w2t = TWO ? reinterpret_cast<const char*>(p.w2b + (size_t)slab * p.nrow_pad * NX) : w2s;
        // x = u + NT q advances by whole column blocks of pass 1 (at most 64 columns, NT >= 64): constant stride
        const unsigned offA = w2_offset(p, kyA, u) * 8u, offB = w2_offset(p, kyB, u) * 8u;
        const unsigned qstr = (unsigned)(((NT >> p.l_cw) * 2) << (p.l_rk + p.l_2gy)) * 8u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const cf*>(w2s + (offA + qstr * (unsigned)q));
            b[q] = *reinterpret_cast<const cf*>(w2t + (offB + qstr * (unsigned)q));
        }
    };

    int un = (int)blockIdx.x;
    if (un >= total) return;
#ifndef XRFT_EMULATE
    // (tuning, XRFTHIP_YTUNE bits 8-15: the k-th resident of a CU -- blocks 256 k .. 256 k + 255 of the launch -- starts k n microseconds late, so
    // that the residents of a CU run their transform and radial-sum phases out of step for the whole launch: the workgroups are persistent)
    for (int i = 0, n = (int)(blockIdx.x >> 8) * ((p.tune >> 8) & 0xff); i < n; ++i) __builtin_amdgcn_s_sleep(32);
#endif
    const int tn = p.tune >> 24;  // (tuning bits 24-27: 1 prefetch in front of the radial sums instead of behind them, 2 no prefetch, 4 no partial stores, 8 no gather)
    if (TIM) tlast = XRFT_ISO_CLOCK();
    if (!(tn & 2)) load_rows(un, (int)threadIdx.x % GX, (int)threadIdx.x / GX);
    for (;;) {
        // (every phase re-derives its lane indices from an opaque copy of the thread index: hoisted out of the persistent loop by the
        // optimiser, the addresses of all phases stayed live through the transforms -- 141 spilled registers)
        int tid = threadIdx.x;
        XRFT_OPAQUE(tid);
        const int g = tid % GX, u = tid / GX;
        cf* mine = lds + g * GSTR;
        const int slab = un / upr, unit = un % upr, ky0 = unit * NRW;
        const int kyA = min(ky0 + g, nyh), kyB = TWO ? kyA : min(ky0 + GX + g, nyh);
        if (tn & 2) load_rows(un, g, u);
        // ---- what the radial sums of THIS unit read from global memory, requested at the top of the unit: the transforms cover the L2 round
        // trip, and the next unit's rows queue BEHIND these (loads return in order: a table load behind the prefetch would wait for all of it)
        int tid2 = threadIdx.x;
        XRFT_OPAQUE(tid2);
        const int row = tid2 % NRW, ky = ky0 + row;
        const bool live = ky <= nyh, twin = ky != 0 && ky != nyh;
        unsigned masks[NSEG];
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int sg = tid2 + s * THR, srow = sg / SPR, s16 = sg % SPR;
            masks[s] = p.tcodes[(size_t)min(ky0 + srow, nyh) * SPR + s16] >> 16;  // bit i: the bin changes between samples i - 1 and i
        }
        // only the bins the unit's rows reach, |k| = ky0 dky .. |(ky0 + NRW - 1, nx/2)| (p.twin, from the map itself)
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
        unsigned rng[KPRE];
#pragma unroll
        for (int k = 0; k < KPRE; ++k) rng[k] = range_of(blo + k * BPI + tid2 / NRW);
        if (addback) {  // add back wx[x] * (subtracted line - plane fit) in the spectral domain (see fasty_cols_kernel)
            const char* __restrict__ crb = reinterpret_cast<const char*>(p.corr + (size_t)slab * NX * 2);
            cf cr[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) cr[q] = *reinterpret_cast<const cf*>(crb + (unsigned)(u + NT * q) * 8u);
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
            if (TWO) {  // the second field has its own residual trend
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
        lap(0, true);  // the rows (and the trend tables) have landed
        cf w1l = w1;  // (opaque: as a loop invariant its fifteen powers were computed once and spilled)
        XRFT_OPAQUE(w1l.re);
        XRFT_OPAQUE(w1l.im);
        fft_p2_pair_w<NX>(a, b, u, mine, w1l, tw2);
        lap(1, false);
        // ---- stage |F|^2 * scale (power) / F0 conj(F1) * scale (cross) in natural order with the conflict-free 17/16 padding
        const int g2 = tid2 % GX, u2 = tid2 / GX;
        if (!TWO) {
#pragma unroll
            for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                for (int k3 = 0; k3 < G::R3; ++k3) {
                    const int sl = nat16(held_k<NX>(u2, bb, k3));
                    const cf va = a[bb * G::R3 + k3], vb = b[bb * G::R3 + k3];
                    stg[g2 * RSP + sl] = (va.re * va.re + va.im * va.im) * p.scale;
                    stg[(GX + g2) * RSP + sl] = (vb.re * vb.re + vb.im * vb.im) * p.scale;
                }
        } else {
#pragma unroll
            for (int bb = 0; bb < G::NB; ++bb)
#pragma unroll
                for (int k3 = 0; k3 < G::R3; ++k3)
                    lds[g2 * RSC + nat16(held_k<NX>(u2, bb, k3))] = cscale(cmulc(a[bb * G::R3 + k3], b[bb * G::R3 + k3]), p.scale);  // xrft.py:825
        }
        __syncthreads();
        // ---- the next unit's rows: in flight during the radial sums
        lap(2, true);  // (profiling build: the table loads are waited for here, in front of the prefetch)
        const int nun = un + (int)gridDim.x;
        const bool more = nun < total;
        if (more && (tn & 1)) {  // (tuning: the prefetch in front of the radial sums -- its 32 loads per wave block the issue for 4000 cycles there: 21.4 against 19.2 us per 4096^2 slab)
            int tid3 = threadIdx.x;
            XRFT_OPAQUE(tid3);
            load_rows(nun, tid3 % GX, tid3 / GX);
        }
        lap(3, false);  // (the prefetch is issued)
        int tid4 = threadIdx.x;
        XRFT_OPAQUE(tid4);
        // (1) every 16-sample segment reduced in place: the sum of each run of equal bins lands on the run's last sample (float32 adds
        // in sample order), all 16 samples first, the running sums in registers, every position written back -- no branch, one trip to the LDS
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int sg = tid4 + s * THR, srow = sg / SPR, s16 = sg % SPR;
            const unsigned mask = masks[s];
            float* q = stg + srow * RSI + CPS * (17 * s16);  // nat16(16 s16) = 17 s16: the segment is contiguous
            float vr[16], vi[TWO ? 16 : 1];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                vr[i] = q[CPS * i];
                if (TWO) vi[TWO ? i : 0] = q[2 * i + 1];
            }
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sr += vr[i];
                vr[i] = sr;
                if (TWO) { si += vi[TWO ? i : 0]; vi[TWO ? i : 0] = si; }
                const bool end = ((mask >> (i + 1)) & 1u) != 0u;
                sr = end ? 0.f : sr;
                si = end ? 0.f : si;
            }
            if (ky0 + srow <= nyh) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    q[CPS * i] = vr[i];
                    if (TWO) q[2 * i + 1] = vi[TWO ? i : 0];
                }
            }
        }
        lap(4, false);
        __syncthreads();
        lap(5, false);  // (what the slowest wave of the workgroup kept this one waiting)
        // (2) the owner of a (bin, row) walks the bin's two kx ranges of that row, one staged value per segment it touches, and adds them in
        // float64 in a fixed order (+kx side by rising segment, then the -kx side); the NRW rows of a bin meet in lane order.  The phase is a
        // chain of LDS latencies, not of work: the first two segments of either side of ALL sweeps are read up front, branch-free (a bin
        // narrower than 16 samples touches no more; the few lanes whose bin hugs |k| = ky walk on in a loop), and the rows of a bin are
        // adjacent lanes of a quad, added by DPP moves instead of trips through the LDS crossbar.
        double* __restrict__ part = p.iso_part + ((size_t)slab * upr + unit) * p.nbins * HW;
        const float* rowp = stg + row * RSI;
        const int blane = tid4 / NRW;
        auto sweeps = [&](const int k0, const int kn, const unsigned* rg) {
            radial_gather_batch<NX, NRW, CPS, BPI, KB>(rowp, row, live, twin, blo, bhi, k0, kn, rg, blane, part, !(tn & 4));
        };
        {
            const int nsw = (tn & 8) ? 0 : (bhi - blo + BPI - 1) / BPI;  // sweeps of this unit (uniform)
#pragma unroll
            for (int b = 0; b < KPRE / KB; ++b)
                if (nsw > b * KB) sweeps(b * KB, min(nsw - b * KB, KB), rng + b * KB);
            for (int k0 = KPRE; k0 < nsw; k0 += KB) {  // (a finer bin map than nfactor = 4: these ranges queue behind the prefetch)
                unsigned rg[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) rg[k] = range_of(blo + (k0 + k) * BPI + blane);
                sweeps(k0, min(nsw - k0, KB), rg);
            }
        }
        if (TIM) {
            const long long t = XRFT_ISO_CLOCK();
            tacc[6] += t - tlast;
            tlast = t;
            tacc[7] += 1;
        }
        if (more && !(tn & 3)) {
            int tid5 = threadIdx.x;
            XRFT_OPAQUE(tid5);
            load_rows(nun, tid5 % GX, tid5 / GX);
        }
        if (!more) break;
        un = nun;
        __syncthreads();  // the next unit's transforms overwrite the staged rows
    }
    if (TIM && threadIdx.x == 0 && p.tim != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.tim[(size_t)blockIdx.x * 8 + i] = tacc[i];
    }
}

}  // namespace xrft

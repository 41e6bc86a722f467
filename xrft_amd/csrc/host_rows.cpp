#include "plan.h"

// one pass over small float32 slabs (fasts.h): resident workgroups walk the slabs
template <int RY, int RX> SGeomRt sgeom_t() {
    typedef SGeom<RY, RX> G;
    const int by_lds = (int)((160 * 1024) / G::LDS);
    return {G::T, G::LDS, std::max(1, std::min(by_lds, (int)G::PER_CU)), G::LDS_ISO};
}
SGeomRt sgeom(long long ny, long long nx) {
#define SG_(A, B) if (ny == 32 * A && nx == 32 * B) return sgeom_t<A, B>();
    SG_(2, 2) SG_(2, 4) SG_(2, 8) SG_(4, 2) SG_(4, 4) SG_(4, 8) SG_(8, 2) SG_(8, 4) SG_(8, 8)
#undef SG_
    return {0, 0, 0, 0};
}
// fasts: is the bin map a radial one (see fasts_power_kernel, ISO)?  If so: first[ky][b] = the smallest |kx| <= nx/2 of row ky whose bin is
// >= b (nx/2 + 1 if none), ky <= ny/2, b = 0 .. nbins.  Otherwise the plan leaves the one-pass path.
int fasts_build_tfirst(xrfthip_plan* P, const int32_t* bm) {
    const int ny = (int)P->d.ny, nx = (int)P->d.nx, nyh = ny / 2, H = nx / 2;
    bool radial = P->nbins <= sgeom(ny, nx).thr && P->nbins >= 1;
    for (int ky = 0; ky <= nyh && radial; ++ky) {
        const int32_t* r = bm + (size_t)ky * nx;
        const bool twin = ky != 0 && 2 * ky != ny;
        const int32_t* t = bm + (size_t)(twin ? ny - ky : ky) * nx;
        for (int m = 0; m <= H; ++m) {
            const int32_t c = r[m];
            if (c < 0 || c >= P->nbins || (m > 0 && c < r[m - 1]) || (m >= 1 && m < H && r[nx - m] != c)) { radial = false; break; }
            if (twin && (t[m] != c || t[(nx - m) % nx] != c)) { radial = false; break; }
        }
    }
    if (!radial) { P->fasts = false; return XRFTHIP_OK; }
    std::vector<uint16_t> f((size_t)(nyh + 1) * (P->nbins + 1), (uint16_t)(H + 1));
    for (int ky = 0; ky <= nyh; ++ky) {
        const int32_t* r = bm + (size_t)ky * nx;
        uint16_t* dst = f.data() + (size_t)ky * (P->nbins + 1);
        int m = 0;
        for (int b = 0; b <= P->nbins; ++b) {
            while (m <= H && r[m] < b) ++m;
            dst[b] = (uint16_t)m;
        }
    }
    return P->s_tfirst.upload(f.data(), f.size() * sizeof(uint16_t));
}

int run_fasts(const xrfthip_plan* P, const void* in, void* out, double* iso, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    FastS p{};
    p.in = (const float*)in; p.out = (float*)out;
    p.tw_y = (const cf*)P->tw_sy.p; p.tw_x = (const cf*)P->tw_sx.p;
    const bool win = P->win[0].p || P->win[1].p;
    p.win_y = win ? (const float*)(P->win[0].p ? P->win[0].p : P->ones4096.p) : nullptr;
    p.win_x = win ? (const float*)(P->win[1].p ? P->win[1].p : P->ones4096.p) : nullptr;
    p.nslabs = d.batch;
    p.detrend = d.detrend;
    p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
    p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
    p.scale = (float)d.scale;
    const SGeomRt G = sgeom(d.ny, d.nx);
    // one workgroup per slab by default: measured against the resident set (kCUs x per_cu workgroups walking the slabs), (16384, 128, 128)
    // linear + Hann 531 vs 425 GFFT/s, (65536, 64, 64) 577 vs 497, 256 x 256 even (profiles/r04_fasts.txt)
    // ... except a 256 x 256 power spectrum (ONE 1024-thread workgroup per CU): a resident set that asks for its next slab while the staged rows of the
    // current one leave (fasts_power_kernel PRE), when every workgroup has several slabs to walk (profiles/r06_fasts_prefetch.txt)
    const bool walk = G.thr >= 1024 && d.out_mode == XRFTHIP_OUT_POWER && d.batch >= 4LL * kCUs * G.per_cu;
    const long long res = P->tune_sgrid < 0 ? (walk ? (long long)kCUs * G.per_cu : 0) : P->tune_sgrid;
    const long long g = res > 0 ? std::min<long long>(res, d.batch) : d.batch;
    p.stagger = (res > 0 && d.batch >= 2 * g) ? (int)P->tune_sstagger : 0;
    const dim3 grid((unsigned)std::min<long long>(g, 0x7fffffffLL)), blk((unsigned)G.thr);
    xrfthip_plan::ProfRec* rec = prof_begin(P, "fasts_slab", st);
    const int isom = (d.flags & XRFTHIP_ISO) ? ((d.flags & XRFTHIP_NO_SPECTRUM_OUT) ? 2 : 1) : 0;
    p.iso = iso; p.tfirst = (const unsigned short*)P->s_tfirst.p; p.nbins = P->nbins;
    const bool cplx = d.out_mode == XRFTHIP_OUT_COMPLEX;
    p.ph_y = (const cf*)P->fph[0].p; p.ph_x = (const cf*)P->fph[1].p; p.ph_on = (cplx && P->fph_on) ? 1 : 0;
#define SL_(A, B) if (d.ny == 32 * A && d.nx == 32 * B) { \
        if (cplx) { auto k = &fasts_power_kernel<A, B, 0, 0>; XRFT_LAUNCH(k, grid, blk, G.lds, st, p); } \
        else if (isom == 0) { auto k = &fasts_power_kernel<A, B, 0>; XRFT_LAUNCH(k, grid, blk, G.lds, st, p); } \
        else if (isom == 1) { auto k = &fasts_power_kernel<A, B, 1>; XRFT_LAUNCH(k, grid, blk, G.lds_iso, st, p); } \
        else { auto k = &fasts_power_kernel<A, B, 2>; XRFT_LAUNCH(k, grid, blk, G.lds_iso, st, p); } }
    SL_(2, 2) SL_(2, 4) SL_(2, 8) SL_(4, 2) SL_(4, 4) SL_(4, 8) SL_(8, 2) SL_(8, 4) SL_(8, 8)
#undef SL_
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

// one pass over 65536-sample float32 rows (fastr.h): a 1024-thread workgroup per row, or a resident set walking the rows
int run_fastr(const xrfthip_plan* P, const void* in, void* out, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    if (P->fastr_rows) {  // complex rows of 256 .. 4096 points: the row pass of the complex two-pass pipeline on the input's own rows
        const bool c2r = (d.flags & XRFTHIP_C2R_X) != 0;
        const long long nxt = c2r ? d.nx / 2 : d.nx;
        const YGeomRt R = yrows_geom(nxt);
        FastYC p{};
        p.c2r = c2r ? 1 : 0; p.in_pitch = (int)(c2r ? nxt + 1 : d.nx); p.w2_nxb = 1;
        p.tw_big = reinterpret_cast<const cf*>(P->tw_big1d.p);
        p.w2 = reinterpret_cast<cf*>(const_cast<void*>(in));
        p.out = out;
        p.tw_x = reinterpret_cast<const cf*>(P->tw_fx.p);
        p.win_y = p.win_x = reinterpret_cast<const float*>(P->win[1].p ? P->win[1].p : P->ones4096.p);
        p.win_on = P->win[1].p ? 1 : 0;
        p.ph_y = p.ph_x = reinterpret_cast<const cf*>(P->fph[1].p);
        const bool phase = d.out_mode == XRFTHIP_OUT_COMPLEX && P->fph_on;
        p.ph_in = (phase && (d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
        p.ph_on = (phase && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
        p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
        p.ishift_x = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_X)) ? 1 : 0;
        p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
        p.ny = R.rk; p.nx = (int)d.nx; p.nslab = 1;  // (ny: one unit of rows -- the kernel addresses by row number)
        p.l_cw = ilog2i((int)nxt); p.l_rk = 0;
        p.power = d.out_mode == XRFTHIP_OUT_POWER ? 1 : 0;
        p.scale = (float)d.scale;
        p.nrows = d.batch;
        xrfthip_plan::ProfRec* rec = prof_begin(P, "fastyc_rows", st);
        const dim3 gridr((unsigned)((d.batch + R.rk - 1) / R.rk)), blkr((unsigned)R.thr);
#define YCR_(NN) do { auto k = &fastyc_rows_kernel<NN>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
#define YC2_(NN) do { auto k = &fastyc_rows_c2r_kernel<NN, true>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
        if (c2r) { if (nxt == 2048) YC2_(2048); else if (nxt == 1024) YC2_(1024); else if (nxt == 512) YC2_(512); else YC2_(256); }
        else if (d.nx == 4096) YCR_(4096); else if (d.nx == 2048) YCR_(2048); else if (d.nx == 1024) YCR_(1024); else if (d.nx == 512) YCR_(512); else YCR_(256);
#undef YCR_
#undef YC2_
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
        return XRFTHIP_OK;
    }
    FastR p{};
    p.in = (const float*)in; p.out = out;
    p.tw_m = (const cf*)P->tw_rm.p; p.tw_s = (const cf*)P->tw_rs.p; p.tw_n = (const cf*)P->tw_rn.p;
    p.win = (const float*)P->win[1].p;
    p.ph = (const cf*)P->fph[1].p; p.ph_on = (d.out_mode == XRFTHIP_OUT_COMPLEX && P->fph_on) ? 1 : 0;
    p.nrows = d.batch;
    p.detrend = d.detrend;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0;
    p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.shift = (d.flags & XRFTHIP_SHIFT_X) ? 1 : 0;
    p.scale = (float)d.scale;
    p.stagger = (int)P->tune_rstagger;
    if (P->fastr_cin) {  // (the flags as fastm_xonly_kernel reads them: the input rotated and conjugated for an inverse, the phase table on the input or on the output)
        p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
        p.ishift = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_X)) ? 1 : 0;
        p.ph_in = ((d.flags & XRFTHIP_PHASE_IN) && P->fph_on) ? 1 : 0;
        p.ph_on = (d.out_mode == XRFTHIP_OUT_COMPLEX && P->fph_on && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
    }
    const long long g = P->tune_rgrid > 0 ? std::min<long long>(P->tune_rgrid, d.batch) : d.batch;
    if (P->tune_rstagger < 0) {
        // rows of 32768 / 16384 samples on a resident set: three classes of workgroups 3.4 us apart for a complex result (6.8 us when the true-phase table rides
        // along: fft (2048, 32768) 275 -> 346 GFFT/s, dft 415 -> 435, fft (4096, 16384) 333 -> 356); a power spectrum gains nothing from a stagger
        const bool cplx = d.out_mode == XRFTHIP_OUT_COMPLEX && !P->fastr_cin;
        p.stagger = d.batch < 2 * g ? 0 : (cplx && d.nx == 32768) ? ((3 << 8) | (p.ph_on ? 2 : 1)) : (cplx && d.nx == 16384) ? ((3 << 8) | 1) : 0;
    }
    const dim3 grid((unsigned)std::min<long long>(g, 0x7fffffffLL)), blk((unsigned)(P->fastr_cin ? d.nx / 32 : d.nx / 64));
    const bool pw = d.out_mode == XRFTHIP_OUT_POWER;
    // profiling (bench.py's roofline.kernel): the start / stop timestamps ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL)
    // instead of two event records around it -- barrier packets either side of a 0.18-ms kernel cost the C2 bench line 50 us per step
    hipEvent_t ea = nullptr, eb = nullptr;
#ifndef XRFT_EMULATE
    if (P->prof && P->prof_recs.size() + 1 < P->prof_recs.capacity() && hipEventCreate(&ea) == hipSuccess) {
        if (hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); ea = nullptr; }
    }
#define RK_(KK, LL) do { auto k = &KK; if (ea) hipExtLaunchKernelGGL(k, grid, blk, LL, st, ea, eb, 0, p); else XRFT_LAUNCH(k, grid, blk, LL, st, p); } while (0)
#else
#define RK_(KK, LL) do { auto k = &KK; XRFT_LAUNCH(k, grid, blk, LL, st, p); } while (0)
#endif
#define RL_(MM, HH) do { \
        if (d.nx == 65536) RK_((fastr_kernel<MM, HH>), kFastRLds); \
        else if (d.nx == 32768) RK_((fastr2_kernel<32, 16, MM, HH>), (R2Geom<32, 16>::LDS)); \
        else if (d.nx == 16384) RK_((fastr2_kernel<16, 16, MM, HH>), (R2Geom<16, 16>::LDS)); \
        else if (d.nx == 8192) RK_((fastr2_kernel<16, 8, MM, HH>), (R2Geom<16, 8>::LDS)); \
        else RK_((fastr2_kernel<8, 8, MM, HH>), (R2Geom<8, 8>::LDS)); } while (0)
#define RC_(MM) do { \
        if (d.nx == 16384) RK_((fastc_kernel<32, 16, MM>), (R2Geom<32, 16>::LDS)); \
        else if (d.nx == 8192) RK_((fastc_kernel<16, 16, MM>), (R2Geom<16, 16>::LDS)); \
        else if (d.nx == 4096) RK_((fastc_kernel<16, 8, MM>), (R2Geom<16, 8>::LDS)); \
        else RK_((fastc_kernel<8, 8, MM>), (R2Geom<8, 8>::LDS)); } while (0)
    if (P->fastr_cin) { if (pw) RC_(1); else RC_(0); }
    else if (pw) { if (p.half) RL_(1, true); else RL_(1, false); } else { if (p.half) RL_(0, true); else RL_(0, false); }
#undef RC_
#undef RL_
#undef RK_
    if (ea) {
        xrfthip_plan::ProfRec r;
        r.label = "fastr_row"; r.a = ea; r.b = eb;
        const_cast<xrfthip_plan*>(P)->prof_recs.push_back(r);
    }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}


// kernels of this unit that take more than 64 KB of dynamic LDS (the register-resident one-pass kernels): called once through set_kernel_attrs_once()
void set_attrs_rows() {
    const int m = (int)kLdsMax;
#define SETF(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    SETF((fasts_power_kernel<8, 8, 0, 0>)); SETF((fasts_power_kernel<8, 4, 0, 0>)); SETF((fasts_power_kernel<4, 8, 0, 0>));
    SETF((fasts_power_kernel<8, 8, 0>)); SETF((fasts_power_kernel<8, 8, 1>)); SETF((fasts_power_kernel<8, 8, 2>));  // (above 64 KB of dynamic LDS)
    SETF((fasts_power_kernel<8, 4, 0>)); SETF((fasts_power_kernel<8, 4, 1>)); SETF((fasts_power_kernel<8, 4, 2>));
    SETF((fasts_power_kernel<4, 8, 0>)); SETF((fasts_power_kernel<4, 8, 1>)); SETF((fasts_power_kernel<4, 8, 2>));
    SETF((fastr_kernel<0, false>)); SETF((fastr_kernel<0, true>)); SETF((fastr_kernel<1, false>)); SETF((fastr_kernel<1, true>));
    SETF((fastc_kernel<32, 16, 0>)); SETF((fastc_kernel<32, 16, 1>)); SETF((fastc_kernel<16, 16, 0>)); SETF((fastc_kernel<16, 16, 1>));
    SETF((fastc_kernel<16, 8, 0>)); SETF((fastc_kernel<16, 8, 1>)); SETF((fastc_kernel<8, 8, 0>)); SETF((fastc_kernel<8, 8, 1>));
    SETF((fastr2_kernel<32, 16, 0, false>)); SETF((fastr2_kernel<32, 16, 0, true>)); SETF((fastr2_kernel<32, 16, 1, false>)); SETF((fastr2_kernel<32, 16, 1, true>));
    SETF((fastr2_kernel<16, 16, 0, false>)); SETF((fastr2_kernel<16, 16, 0, true>)); SETF((fastr2_kernel<16, 16, 1, false>)); SETF((fastr2_kernel<16, 16, 1, true>));
#undef SETF
}

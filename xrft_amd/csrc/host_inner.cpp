#include "plan.h"

// ---------------------------------------------------------------------------------------------------------------
// xrfthip_desc.inner > 1 as two fused passes (fastn.h, round 5): 16 bytes per sample through memory where the composite of two one-axis plans moves 32
// ---------------------------------------------------------------------------------------------------------------
int fusedi_tables(xrfthip_plan* P) {  // what depends on the windows / phases: called from finalize_plan
    const xrfthip_desc& d = P->d;
    const long long ne = std::max<long long>(d.inner, std::max<long long>(d.mid, 1)), ncol = d.nx * ne;
    const long long sx = d.mid > 1 ? 1 : d.inner, se = d.mid > 1 ? d.nx : 1;
    std::vector<double> wexp((size_t)ncol);
    for (long long x = 0; x < d.nx; ++x) {
        const double w = P->host_win_x.empty() ? 1.0 : P->host_win_x[(size_t)x];
        for (long long e = 0; e < ne; ++e) wexp[(size_t)(x * sx + e * se)] = w;
    }
    int rc = upload_real_table(P, P->winx_exp, wexp.data(), ncol, 0);
    if (!rc) rc = fasty_window_spectra(P);
    if (!rc && d.out_mode != XRFTHIP_OUT_POWER) rc = fast_phase_tables(P);
    return rc;
}

xrfthip_plan* create_fused_inner(const xrfthip_desc& d) {
    if (env_ll("XRFTHIP_NO_FAST", 0) || !env_ll("XRFTHIP_FASTN", 1) || !env_ll("XRFTHIP_FUSED_INNER", 1)) return nullptr;
    // (the independent elements innermost, or between the two axes -- not both)
    if (d.ndim != 2 || !((d.mid <= 1 && d.inner >= 2) || (d.mid >= 2 && d.inner <= 1)) || (d.dtype != XRFTHIP_F32 && d.dtype != XRFTHIP_F64)) return nullptr;
    const bool midlay = d.mid >= 2;
    const long long ne = midlay ? d.mid : d.inner;
    // (the cross spectrum of two real fields, round 6: pass 1 and the plane fit once per field, pass 2 on GE / 2 elements of both)
    const bool crossm = d.out_mode == XRFTHIP_OUT_CROSS;
    if (d.out_mode != XRFTHIP_OUT_COMPLEX && d.out_mode != XRFTHIP_OUT_POWER && !crossm) return nullptr;
    // (real_dim along the second axis -- HALF_X, the power spectrum's REALDIM_X2: rows of nx/2 + 1 samples out of pass 2, unshifted along x)
    const uint32_t ok = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | XRFTHIP_HALF_X | XRFTHIP_HALF_Y | (d.out_mode != XRFTHIP_OUT_POWER ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X) : 0u) |
                        (d.out_mode != XRFTHIP_OUT_COMPLEX ? XRFTHIP_REALDIM_X2 : 0u);
    if (d.flags & ~ok) return nullptr;  // (a flipped axis: the composite of one-axis plans)
    if ((d.flags & XRFTHIP_HALF_X) && ((d.flags & XRFTHIP_SHIFT_X) || (d.nx & 1))) return nullptr;
    if ((d.flags & XRFTHIP_REALDIM_X2) && !(d.flags & (XRFTHIP_HALF_X | XRFTHIP_HALF_Y))) return nullptr;
    const bool dbl = d.dtype == XRFTHIP_F64;
    const size_t rs = dbl ? 8 : 4, cs = 2 * rs;
    const long long ncol = d.nx * ne;
    if (d.ny < 16 || d.nx < 16 || d.ny > 8192 || d.nx > 8192 || ncol > (1LL << 26) || (unsigned long long)d.ny * (unsigned long long)ncol * rs >= (1ULL << 32)) return nullptr;
    const int maxthr = dbl ? fastn_max_threads<double>() : fastn_max_threads<float>();
    const int maxr = dbl ? fastn_max_radix<double>() : fastn_max_radix<float>();
    // pass 1: ny-point columns of the [ny][nx inner] view, the widest column blocks that leave three, two, one workgroup on a CU
    NGeo gc{}, gr{};
    int G = 0, GE = 0;
    // Both passes here are "column" passes (the sequences of pass 2 lie `inner` apart): the widest block of sequences that leaves two workgroups on a CU, then one --
    // 64-byte pieces of every line where the LDS allows ((1024, 1024, 64) float32: 8 column pairs 187 us against 273 with 4 and three workgroups; 8 elements
    // per row workgroup 231 us against 290; profiles/r05_inner_knobs.txt)
    static const size_t caps[] = {78 * 1024, 156 * 1024, 156 * 1024};
    const int f_gc = (int)env_ll("XRFTHIP_FI_GC", 0), f_ge = (int)env_ll("XRFTHIP_FI_GE", 0), f_tc = (int)env_ll("XRFTHIP_FI_TC", 0), f_tr = (int)env_ll("XRFTHIP_FI_TR", 0);  // (measurements)
    for (int ci = 0; ci < 3 && !G; ++ci)
        for (int cand = f_gc ? f_gc : (dbl ? 4 : 8); cand >= 1 && !G; cand >>= 1) {
            NGeo t{};
            // (float64: 256 threads -- (1024, 1024, 32): 176 us against 235 with the 384 that keep the most waves resident)
            const long long pts = (long long)cand * d.ny;  // (short columns: fastn_setup's rule)
            const int tc = f_tc ? f_tc : pts <= (dbl ? 768 : 1536) ? 64 : dbl ? (pts <= 2048 ? 128 : 256) : pts < 4096 ? 256 : 0;
            bool picked = false;
            for (int tt = tc; tt && tt <= 256 && !picked && !f_tc; tt *= 2) picked = fastn_pick(d.ny, cand, false, dbl, true, maxr, tt, t);
            if (!picked && !fastn_pick(d.ny, cand, false, dbl, true, maxr, f_tc, t)) continue;
            if ((long long)t.g * (d.ny / t.r[t.np - 1]) > maxthr) continue;
            if (fastn_lds(t, cs, true) <= (f_gc ? caps[2] : caps[ci])) { G = cand; gc = t; }
        }
    // ... or, a length with ONE prime 17 ... 127 (1460 = 20 x 73 six-hourly samples of a year along "time"): the Rader columns (fastn_cols_kernel<T, 2, 16>), as fastn_setup
    int rad_p = 0;
    std::vector<int> rad_rq, rad_rp;
    if (!G && !f_gc && d.ny <= 8192 && env_ll("XRFTHIP_FASTN_RADER", 1) && rader_split(d.ny, true, rad_p, rad_rq, rad_rp)) {
        for (int cand = dbl ? 4 : 8; cand >= 1 && !G; cand >>= 1) {
            if (cand > 1 && (long long)cand * d.ny > 6000) continue;
            NGeo c{};
            c.n = (int)d.ny; c.np = 0; c.g = cand; c.lg = ilog2i(cand); c.str = (int)d.ny; c.twn = (int)(d.ny / rad_p) + rad_p - 1;
            const long long pts = (long long)cand * d.ny;
            c.thr = pts <= 1536 ? 64 : pts <= 2560 ? 128 : (pts >= 4096 && !dbl) ? 512 : 256;
            if (fastn_lds(c, cs, true) + 2 * (((size_t)d.ny + 7) & ~(size_t)7) * 2 <= 156 * 1024) { G = cand; gc = c; }
        }
        if (!G) rad_p = 0;
    }
    // pass 2: GE sequences (consecutive inner elements) of nx points: 64-byte runs of the result where the LDS allows
    // (midlay: the sequences are contiguous rows -- as many as make ~4096 (float64: 2048) points, two at least: (1440, 73, 144) with two 144-point rows per workgroup
    // ran 26 000 tiny workgroups, 115 us)
    // (the elements innermost: 8 (float64: 4) -- 64-byte pieces; SHORT sequences, 16 of them while that is <= 4096 points: (73, 144, 1460) float32 rows 51 -> 42 us,
    // (72, 144, 1440) float64 116 -> 71, (256, 256, 512) 102 -> 73; at 512 points 16 elements were slower than 8.  profiles/r05_inner_small.txt)
    const int ge_in = (16LL * d.nx <= 4096) ? 16 : (dbl ? 4 : 8);
    int ge_mid = 2;
    while (ge_mid < 32 && (long long)ge_mid * 2 * d.nx <= (dbl ? 2048 : 4096)) ge_mid *= 2;
    for (int ci = 0; ci < 3 && !GE; ++ci)
        for (int cand = f_ge ? f_ge : midlay ? ge_mid : ge_in; cand >= 1 && !GE; cand >>= 1) {
            if (cand > 2 * ne) continue;
            NGeo t{};
            // (float32: 512 threads where a workgroup holds eight sequences, 231 us against 292 with 256; float64: 256, 182 us against 249 with 384)
            const int tr = f_tr ? f_tr : dbl ? 256 : ((long long)cand * d.nx > 4096) ? 512 : 0;
            if (!(tr && fastn_pick(d.nx, cand, false, dbl, false, maxr, tr, t)) && !fastn_pick(d.nx, cand, false, dbl, false, maxr, 0, t)) continue;
            if ((long long)t.g * (d.nx / t.r[t.np - 1]) > maxthr) continue;
            if (fastn_lds(t, cs, false) <= (f_ge ? caps[2] : caps[ci])) { GE = cand; gr = t; }
        }
    if (!G || !GE) return nullptr;
    if (crossm && GE < 2) return nullptr;  // (both fields of an element share a row workgroup)
    xrfthip_plan* P = new (std::nothrow) xrfthip_plan();
    if (!P) return nullptr;
    P->d = d;
    P->fusedi = true;
    P->inner = d.inner > 1 ? d.inner : 1; P->mid = midlay ? d.mid : 1;
    P->dbl = dbl; P->cplx_in = false; P->rsize = rs; P->csize = cs;
    P->nx_out = (d.flags & XRFTHIP_HALF_X) ? d.nx / 2 + 1 : d.nx;
    P->yny = d.ny; P->ynx = ncol;  // (the view pass 1 transforms)
    P->n_c.rt = true; P->n_c.geo = gc; P->n_c.lds = fastn_lds(gc, cs, true) + (rad_p ? 2 * (((size_t)d.ny + 7) & ~(size_t)7) * 2 : 0);
    P->n_rad_p = rad_p; P->n_rq = rad_rq; P->n_rp = rad_rp;
    P->n_dbg = (int)env_ll("XRFTHIP_FASTN_DBG", 0); P->fi_dbg = (int)env_ll("XRFTHIP_FI_DBG", 0); P->fi_vec = env_ll("XRFTHIP_FI_VEC", 1) ? 1 : 0;
    P->n_r.rt = true; P->n_r.geo = gr; P->n_r.lds = fastn_lds(gr, cs, false);
    P->n_cw = 2 * G; P->n_nxb = (int)((ncol + P->n_cw - 1) / P->n_cw); P->y_pitch = (long long)P->n_nxb * P->n_cw;
    P->n_rk = (int)std::max<long long>(1, (long long)(128 / (P->n_cw * cs)));
    P->n_rpu = 1;
    P->y_nrow_pad = (int)((d.ny / 2 + 1 + P->n_rk - 1) / P->n_rk * P->n_rk);
    int rc = dbl ? build_twiddle<double>(P->tw_fx, d.nx, d.nx) : build_twiddle<float>(P->tw_fx, d.nx, d.nx);
    if (!rc) rc = dbl ? build_twiddle<double>(P->tw_fy, d.ny, d.ny) : build_twiddle<float>(P->tw_fy, d.ny, d.ny);
    std::vector<double> ones((size_t)std::max<long long>(d.ny, ncol), 1.0);
    if (!rc) rc = upload_real_table(P, P->ones4096, ones.data(), (int64_t)ones.size(), 0);
    if (!rc && rad_p) rc = dbl ? fastn_rader_tables<double>(P) : fastn_rader_tables<float>(P);
    else if (!rc) rc = dbl ? fastn_upload_twm<double>(gc, P->n_c.twm) : fastn_upload_twm<float>(gc, P->n_c.twm);
    if (!rc) rc = dbl ? fastn_upload_twm<double>(gr, P->n_r.twm) : fastn_upload_twm<float>(gr, P->n_r.twm);
    if (!rc) rc = P->n_c.geo_dev.upload(&P->n_c.geo, sizeof(NGeo));
    if (!rc) rc = P->n_r.geo_dev.upload(&P->n_r.geo, sizeof(NGeo));
    if (!rc) rc = fusedi_tables(P);
    if (rc) { delete P; return nullptr; }
    // workspace: the intermediate of one group of slabs, the column sums, the plane corrections
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t slab_w = (size_t)P->y_nrow_pad * (size_t)P->y_pitch * cs;
    long long Gs = d.slabs_per_group > 0 ? d.slabs_per_group : (long long)std::max<size_t>(1, ((size_t)512 << 20) / std::max<size_t>(slab_w * (crossm ? 2 : 1), 1));
    Gs = std::max<long long>(1, std::min<long long>(Gs, std::max<long long>(d.batch, 1)));
    P->G = (int)Gs;
    size_t off = 0;
    const size_t nf = crossm ? 2 : 1;  // fields: field 1's intermediate, sums and corrections lie behind field 0's
    P->off_w = off; off = al(off + nf * (size_t)Gs * slab_w);
    P->off_rowfit = off; off = al(off + nf * (size_t)Gs * ncol * 4 * sizeof(double));
    P->off_corr = off; off = al(off + nf * (size_t)Gs * ncol * cs);
    P->ws_bytes = off;
    return P;
}

int run_fused_inner(const xrfthip_plan* P, const void* in, const void* in1, void* out, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const bool midlay = d.mid >= 2, crossm = d.out_mode == XRFTHIP_OUT_CROSS;
    const long long ne = midlay ? d.mid : d.inner, ncol = d.nx * ne;
    const int sx = midlay ? 1 : (int)d.inner, se = midlay ? (int)d.nx : 1;
    const size_t out_esz = d.out_mode == XRFTHIP_OUT_POWER ? P->rsize : P->csize;
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        const size_t slab_w = (size_t)P->y_nrow_pad * (size_t)P->y_pitch * P->csize;
        const size_t fw = (size_t)P->G * slab_w, ff = (size_t)P->G * ncol * 4 * sizeof(double), fc = (size_t)P->G * ncol * P->csize;  // field 1 behind field 0 (create_fused_inner)
        FastM m{};
        xrfthip_plan::ProfRec* rec = nullptr;
      for (int f = 0; f < (crossm ? 2 : 1); ++f) {
        m = FastM{};
        m.in = (const char*)(f ? in1 : in) + (size_t)g0 * d.ny * ncol * P->rsize;
        m.w2 = ws + P->off_w + f * fw;
        m.tw_y = P->tw_fy.p;
        m.win_y = P->win[0].p ? P->win[0].p : P->ones4096.p;
        m.win_x = P->winx_exp.p;
        m.colfit = reinterpret_cast<double*>(ws + P->off_rowfit + f * ff);
        m.ny = (int)d.ny; m.nx = (int)ncol; m.nrow_pad = P->y_nrow_pad;
        m.l_cw = ilog2i(P->n_cw); m.l_rk = ilog2i(P->n_rk);
        m.detrend = d.detrend; m.nslab = (int)gc; m.nunits = (int)(gc * P->n_nxb);
        rec = prof_begin(P, "fastn_cols", st);
        fastn_launch_cols(P, m, st);
        prof_end(rec, st);
        if (d.detrend) {
            rec = prof_begin(P, "fastn_fit_inner", st);
            const dim3 grid((unsigned)(gc * ne)), blk(256);
            if (P->dbl) { auto k = &fastn_fit_inner_kernel<double>; XRFT_LAUNCH(k, grid, blk, 3 * 256 * sizeof(double), st, (const double*)m.colfit, (const double*)m.win_x, reinterpret_cast<C2<double>*>(ws + P->off_corr + f * fc), (int)d.nx, (int)ne, (int)d.ny, (int)d.detrend, sx, se); }
            else { auto k = &fastn_fit_inner_kernel<float>; XRFT_LAUNCH(k, grid, blk, 3 * 256 * sizeof(double), st, (const double*)m.colfit, (const float*)m.win_x, reinterpret_cast<C2<float>*>(ws + P->off_corr + f * fc), (int)d.nx, (int)ne, (int)d.ny, (int)d.detrend, sx, se); }
            prof_end(rec, st);
        }
      }
        FastNI r{};
        r.w2 = ws + P->off_w; r.corr = ws + P->off_corr; r.what0 = P->ywhat0.p; r.what1 = P->ywhat1.p;
        if (crossm) { r.w2b = ws + P->off_w + fw; r.corrb = ws + P->off_corr + fc; }
        r.tw_x = P->tw_fx.p; r.twm = P->n_r.twm.p; r.g = (NGeoPtr)P->n_r.geo_dev.p;
        r.ph_y = P->fph[0].p; r.ph_x = P->fph[1].p; r.ph_on = (d.out_mode != XRFTHIP_OUT_POWER && P->fph_on) ? 1 : 0;
        r.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0; r.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
        r.half_y = (d.flags & XRFTHIP_HALF_Y) ? 1 : 0;  // (real_dim along the first axis: rows ky = 0 .. ny/2, no twin rows)
        r.out = (char*)out + (size_t)g0 * (size_t)(r.half_y ? d.ny / 2 + 1 : d.ny) * (size_t)P->nx_out * ne * out_esz;
        const int geo = crossm ? P->n_r.geo.g / 2 : P->n_r.geo.g;  // elements a row workgroup writes
        r.ny = (int)d.ny; r.nx = (int)d.nx; r.inner = (int)ne; r.sx = sx; r.se = se; r.midlay = midlay ? 1 : 0; r.nrow_pad = P->y_nrow_pad; r.pitch = (int)P->y_pitch;
        r.l_cw = m.l_cw; r.l_rk = m.l_rk; r.detrend = d.detrend;
        r.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
        r.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
        const NGeo& hg = P->n_r.geo;
        {
            const int vw = (int)(16 / out_esz);
            r.vec = (!midlay && (vw == 1 || (d.inner % vw == 0 && geo % vw == 0))) ? 1 : 0;
            if (!P->fi_vec) r.vec = 0;
            r.dbg = P->fi_dbg;
        }
        r.neb = (int)((ne + geo - 1) / geo);
        r.nunits = (int)(gc * (d.ny / 2 + 1) * r.neb);
        r.scale = d.scale;
        int maxrad = 0;
        for (int i = 0; i < hg.np; ++i) maxrad = std::max(maxrad, hg.r[i]);
        const dim3 grid((unsigned)(8 * ((r.nunits + 7) / 8))), blk((unsigned)hg.thr);
        rec = prof_begin(P, "fastn_irows", st);
#define NI_(TT, CC) do { if (d.out_mode == XRFTHIP_OUT_POWER) { auto k = &fastn_irows_kernel<TT, 1, CC>; XRFT_LAUNCH(k, grid, blk, P->n_r.lds, st, r); } \
                         else if (crossm) { auto k = &fastn_irows_kernel<TT, 2, CC>; XRFT_LAUNCH(k, grid, blk, P->n_r.lds, st, r); } \
                         else { auto k = &fastn_irows_kernel<TT, 0, CC>; XRFT_LAUNCH(k, grid, blk, P->n_r.lds, st, r); } } while (0)
        if (P->dbl) NI_(double, 16); else if (maxrad > 16) NI_(float, 20); else NI_(float, 16);
#undef NI_
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

// composite plan for xrfthip_desc.inner > 1 (see xrfthip_plan::inner)
int create_inner_plan(xrfthip_plan** plan, const xrfthip_desc& d) {
    if (xrfthip_plan* F = create_fused_inner(d)) { *plan = F; return XRFTHIP_OK; }
    if ((d.flags & (XRFTHIP_HALF_X | XRFTHIP_HALF_Y | XRFTHIP_REALDIM_X2)) || d.out_mode == XRFTHIP_OUT_CROSS) return XRFTHIP_UNSUPPORTED_LENGTH;  // (real_dim, two fields: the fused passes only; the caller transposes)
    const uint32_t ok = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X | XRFTHIP_FLIP_Y | XRFTHIP_FLIP_X;
    if (d.ndim != 2 || (d.flags & ~ok) || (d.out_mode != XRFTHIP_OUT_COMPLEX && d.out_mode != XRFTHIP_OUT_POWER)) return XRFTHIP_BAD_ARG;
    if (d.inner > (1LL << 30) || d.mid > (1LL << 30) || d.nx * d.inner > (1LL << 30) || d.mid * d.nx * d.inner > (1LL << 30) || d.batch * d.mid > (1LL << 40)) return XRFTHIP_BAD_ARG;
    xrfthip_plan* P = new (std::nothrow) xrfthip_plan();
    if (!P) return XRFTHIP_ALLOC_FAILED;
    P->d = d;
    P->inner = d.inner; P->mid = d.mid;
    P->dbl = d.dtype == XRFTHIP_F64 || d.dtype == XRFTHIP_C128;
    P->cplx_in = d.dtype >= XRFTHIP_C64;
    P->rsize = P->dbl ? 8 : 4;
    P->csize = 2 * P->rsize;
    P->nx_out = d.nx;
    xrfthip_desc dx = d, dy = d;
    dx.inner = dy.inner = 1; dx.mid = dy.mid = 1;
    dx.detrend = dy.detrend = XRFTHIP_DETREND_NONE;
    dx.out_mode = XRFTHIP_OUT_COMPLEX; dx.scale = 1.0;
    if (d.inner > 1) {
        // x where it lies: [batch ny mid][nx][inner], the per-axis flags of x become the y flags of the one-axis plan
        dx.batch = d.batch * d.ny * d.mid; dx.ny = d.nx; dx.nx = d.inner;
        dx.flags = XRFTHIP_AXIS_Y | ((d.flags & XRFTHIP_SHIFT_X) ? XRFTHIP_SHIFT_Y : 0u) | ((d.flags & XRFTHIP_ISHIFT_X) ? XRFTHIP_ISHIFT_Y : 0u) |
                   ((d.flags & XRFTHIP_FLIP_X) ? XRFTHIP_FLIP_Y : 0u);
    } else {
        // nothing behind x (two transform axes with `mid` elements between them, x the contiguous one): a 1-D plan over the rows [batch ny mid][nx]
        P->sub_x_1d = true;
        dx.ndim = 1; dx.batch = d.batch * d.ny * d.mid; dx.ny = 1; dx.nx = d.nx;
        dx.flags = d.flags & (XRFTHIP_SHIFT_X | XRFTHIP_ISHIFT_X | XRFTHIP_FLIP_X);
    }
    // then y: [batch][ny][mid nx inner], complex input, the requested result and scale
    dy.batch = d.batch; dy.ny = d.ny; dy.nx = d.mid * d.nx * d.inner;
    dy.dtype = P->dbl ? XRFTHIP_C128 : XRFTHIP_C64;
    dy.flags = XRFTHIP_AXIS_Y | (d.flags & (XRFTHIP_SHIFT_Y | XRFTHIP_ISHIFT_Y | XRFTHIP_FLIP_Y));
    int rc = xrfthip_plan_create(&P->sub_x, &dx);
    if (!rc) rc = xrfthip_plan_create(&P->sub_y, &dy);
    if (rc) { delete P; return rc; }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t pts = (size_t)d.batch * d.ny * d.mid * d.nx * d.inner;
    size_t off = 0;
    P->off_sub = off; off = al(off + std::max(P->sub_x->ws_bytes, P->sub_y->ws_bytes));
    P->off_mid = off; off = al(off + pts * P->csize);                                        // the x-transformed field (complex)
    if (d.detrend) {
        P->off_det = off; off = al(off + pts * (P->cplx_in ? P->csize : P->rsize));         // the detrended copy of the input
        P->off_dws = off; off = al(off + detrend_inner_ws(P->cplx_in, d.batch * d.mid, d.inner));
    }
    P->ws_bytes = off;
    *plan = P;
    return XRFTHIP_OK;
}

int run_inner_plan(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const void* cur = in;
    if (d.detrend) {
        int rc = run_detrend_inner(d.dtype, 2, d.batch * d.mid, d.ny, d.nx, d.inner, d.detrend, in, ws + P->off_det, ws + P->off_dws, st, d.mid);
        if (rc) return rc;
        cur = ws + P->off_det;
    }
    int rc = xrfthip_exec(P->sub_x, cur, nullptr, ws + P->off_mid, nullptr, ws + P->off_sub, P->sub_x->ws_bytes, st);
    if (!rc) rc = xrfthip_exec(P->sub_y, ws + P->off_mid, nullptr, out, nullptr, ws + P->off_sub, P->sub_y->ws_bytes, st);
    return rc;
}


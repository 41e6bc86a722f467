#include "plan.h"

// combined per-axis factors of the complex modes: the true-phase table (or 1) times (-1)^k for an ifftshifted input
// (rolling the input by n/2 -- xrft.py:436-441 -- is that sign in the spectrum; in a cross spectrum the two signs cancel)
int fast_phase_tables(xrfthip_plan* P) {
    const xrfthip_desc& d = P->d;
    P->fph_on = false;
    for (int ax = 0; ax < 2; ++ax) {
        const long long n = ax == 0 ? d.ny : d.nx;
        const bool sign = d.out_mode == XRFTHIP_OUT_COMPLEX && !(d.flags & XRFTHIP_INVERSE) && (d.flags & (ax == 0 ? XRFTHIP_ISHIFT_Y : XRFTHIP_ISHIFT_X));  // (an inverse plan rotates its input)
        std::vector<cf> t((size_t)n);
        const bool dtab = (P->fastm || P->fastmy || P->fastmx || P->fastg || P->fastgy || P->fusedi) && P->dbl;
        std::vector<C2<double>> td(dtab ? (size_t)n : 0);
        for (long long k = 0; k < n; ++k) {
            double re = 1.0, im = 0.0;
            if ((size_t)(2 * k + 1) < P->host_phase[ax].size()) { re = P->host_phase[ax][(size_t)(2 * k)]; im = P->host_phase[ax][(size_t)(2 * k + 1)]; }  // (a C2R_X plan: nx/2 + 1 entries on axis 1)
            if (sign && !(n & 1)) { if (k & 1) { re = -re; im = -im; } }
            else if (sign) {  // an odd length (fastg.h takes them): the ifftshift is a rotation by n // 2 samples, X'[k] = X[k] exp(+2 pi i (n // 2) k / n)
                const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)((k * (n / 2)) % n) / (long double)n;
                const double cr = (double)cosl(a), ci = (double)sinl(a), r2 = re * cr - im * ci, i2 = re * ci + im * cr;
                re = r2; im = i2;
            }
            t[(size_t)k].re = (float)re; t[(size_t)k].im = (float)im;
            if (dtab) { td[(size_t)k].re = re; td[(size_t)k].im = im; }
            if (dtab ? (re != 1.0 || im != 0.0) : (t[(size_t)k].re != 1.0f || t[(size_t)k].im != 0.0f)) P->fph_on = true;
        }
        int rc = dtab ? P->fph[ax].upload(td.data(), td.size() * sizeof(C2<double>)) : P->fph[ax].upload(t.data(), t.size() * sizeof(cf));
        if (rc) return rc;
    }
    return XRFTHIP_OK;
}

bool phase_nontrivial(const xrfthip_plan* P) {
    for (int ax = 0; ax < 2; ++ax)
        for (size_t k = 0; k + 1 < P->host_phase[ax].size(); k += 2)
            if (std::fabs(P->host_phase[ax][k] - 1.0) > 1e-12 || std::fabs(P->host_phase[ax][k + 1]) > 1e-12) return true;
    return false;
}

// the specialised path is taken unless an isotropic cross spectrum carries a true-phase factor that is not 1 (two
// fields with different lags): its radial sums would need the factor per sample inside the column pass
bool fast_on(const xrfthip_plan* P) {
    if (P->fastm) return true;
    if (P->fast1d) return true;  // (a window rides on a slab-shaped table: fasty_window_spectra_1d)
    if (!P->fast4096) return false;
    if (P->d.out_mode == XRFTHIP_OUT_CROSS && (P->d.flags & XRFTHIP_ISO) && phase_nontrivial(P)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// two-pass "y first" pipeline (fasty.h): full float32 power spectra of power-of-two slabs
// ---------------------------------------------------------------------------------------------------------------
template <int NY> YGeomRt ycols_geom_t() {
    typedef YCols<NY> Y;
    return {Y::THR, Y::GY, Y::CW, Y::RK, Y::LBS, (size_t)(Y::GY * YLds<NY, Y::GY>::GSTR + 16 * P2<NY>::R3) * sizeof(cf) + (size_t)(Y::THR / 64) * Y::GY * 8 * sizeof(double)};
}
template <int NX, bool FS = false> YGeomRt yrows_geom_t() {
    typedef YRows<NX, FS> R;
    return {R::THR, R::GX, 0, R::RPU, 0, (size_t)(R::GX * YLds<NX, R::GX>::GSTR + 16 * P2<NX>::R3) * sizeof(cf)};
}
int ycols_gstr(long long ny) {  // complex elements of LDS per packed column pair of pass 1 (YLds<NY, GY>::GSTR)
    switch (ny) { case 4096: return YLds<4096, YCols<4096>::GY>::GSTR; case 2048: return YLds<2048, YCols<2048>::GY>::GSTR; case 1024: return YLds<1024, YCols<1024>::GY>::GSTR;
                  case 512: return YLds<512, YCols<512>::GY>::GSTR; default: return YLds<256, YCols<256>::GY>::GSTR; }
}
YGeomRt ycols_geom(long long ny) {
    switch (ny) { case 4096: return ycols_geom_t<4096>(); case 2048: return ycols_geom_t<2048>(); case 1024: return ycols_geom_t<1024>();
                  case 512: return ycols_geom_t<512>(); default: return ycols_geom_t<256>(); }
}
YGeomRt yrows_geom(long long nx, bool fs) {  // .rk = rows per workgroup; fs: the four-step 1-D form (256-point rows)
    if (fs) return yrows_geom_t<256, true>();
    switch (nx) { case 4096: return yrows_geom_t<4096>(); case 2048: return yrows_geom_t<2048>(); case 1024: return yrows_geom_t<1024>();
                  case 512: return yrows_geom_t<512>(); default: return yrows_geom_t<256>(); }
}
long long fasty_rows_gx(const xrfthip_plan* P) { return yrows_geom(P->ynx, P->fast1d).gxy; }
int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Four-step 1-D with a window w[n], n = nx i1 + i2: the slab-shaped float32 window table pass 1 reads, and -- column i2 of the view has
// its own window w[nx i1 + i2] -- the per-column transforms FFT_i1(w) and FFT_i1(w (i1 - ibar)) as tables [i2][k1 < nrow_pad] that carry
// the residual line back in pass 2 (fasty_rows_kernel, W2D).  Host, once per plan: nx transforms of ny points each.
int fasty_window_spectra_1d(xrfthip_plan* P) {
    const int ny = (int)P->yny, nx = (int)P->ynx, nyh = ny / 2, nent = P->y_nrow_pad;
    const std::vector<double>& w = P->host_win_x;
    if ((long long)w.size() != (long long)ny * nx) return XRFTHIP_BAD_ARG;
    std::vector<float> wf(w.size());
    for (size_t i = 0; i < w.size(); ++i) wf[i] = (float)w[i];
    int rc = P->win2d.upload(wf.data(), wf.size() * sizeof(float));
    if (rc) return rc;
    std::vector<cf> h0((size_t)nx * nent), h1((size_t)nx * nent);
    std::vector<double> r0((size_t)ny), i0((size_t)ny), r1((size_t)ny), i1((size_t)ny);
    for (int x = 0; x < nx; ++x) {
        for (int i = 0; i < ny; ++i) {
            const double wv = w[(size_t)i * nx + x];
            r0[(size_t)i] = wv; i0[(size_t)i] = 0.0;
            r1[(size_t)i] = wv * ((double)i - 0.5 * (ny - 1)); i1[(size_t)i] = 0.0;
        }
        host_fft_pow2(r0, i0);
        host_fft_pow2(r1, i1);
        for (int k = 0; k < nent; ++k) {
            cf a, b;
            a.re = k <= nyh ? (float)r0[(size_t)k] : 0.f; a.im = k <= nyh ? (float)i0[(size_t)k] : 0.f;
            b.re = k <= nyh ? (float)r1[(size_t)k] : 0.f; b.im = k <= nyh ? (float)i1[(size_t)k] : 0.f;
            h0[(size_t)x * nent + k] = a; h1[(size_t)x * nent + k] = b;
        }
    }
    rc = P->ywhat0.upload(h0.data(), h0.size() * sizeof(cf));
    if (!rc) rc = P->ywhat1.upload(h1.data(), h1.size() * sizeof(cf));
    return rc;
}

// FFT_y(wy) and FFT_y(wy (i - ibar)) for ky < nrow_pad (zero beyond ny/2): what pass 2 needs to add the residual trend back
int fasty_window_spectra(xrfthip_plan* P) {
    const int ny = (int)P->yny, nyh = ny / 2;
    std::vector<double> r0((size_t)ny), i0((size_t)ny, 0.0), r1((size_t)ny), i1((size_t)ny, 0.0);
    for (int i = 0; i < ny; ++i) {
        const double w = P->host_win_y.empty() ? 1.0 : P->host_win_y[(size_t)i];
        r0[(size_t)i] = w;
        r1[(size_t)i] = w * ((double)i - 0.5 * (ny - 1));
    }
    if ((ny & (ny - 1)) == 0) { host_fft_pow2(r0, i0); host_fft_pow2(r1, i1); }
    else {  // a direct O(n^2) transform with exact twiddle indices (n <= 1440, once per plan)
        std::vector<long double> cw((size_t)ny), sw((size_t)ny);
        for (int k = 0; k < ny; ++k) { const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)ny; cw[(size_t)k] = cosl(a); sw[(size_t)k] = sinl(a); }
        std::vector<double> o0r((size_t)ny), o0i((size_t)ny), o1r((size_t)ny), o1i((size_t)ny);
        for (int k = 0; k <= nyh; ++k) {
            long double a0 = 0, b0 = 0, a1 = 0, b1 = 0;
            for (int i = 0; i < ny; ++i) {
                const size_t m = (size_t)(((long long)i * k) % ny);
                a0 += r0[(size_t)i] * cw[m]; b0 += r0[(size_t)i] * sw[m];
                a1 += r1[(size_t)i] * cw[m]; b1 += r1[(size_t)i] * sw[m];
            }
            o0r[(size_t)k] = (double)a0; o0i[(size_t)k] = (double)b0; o1r[(size_t)k] = (double)a1; o1i[(size_t)k] = (double)b1;
        }
        r0 = o0r; i0 = o0i; r1 = o1r; i1 = o1i;
    }
    const int nent = P->y_nrow_pad;
    if ((P->fastm || P->fusedi) && P->dbl) {
        std::vector<C2<double>> d0((size_t)nent), d1((size_t)nent);
        for (int k = 0; k < nent; ++k) {
            d0[(size_t)k].re = k <= nyh ? r0[(size_t)k] : 0.0; d0[(size_t)k].im = k <= nyh ? i0[(size_t)k] : 0.0;
            d1[(size_t)k].re = k <= nyh ? r1[(size_t)k] : 0.0; d1[(size_t)k].im = k <= nyh ? i1[(size_t)k] : 0.0;
        }
        int rcd = P->ywhat0.upload(d0.data(), d0.size() * sizeof(C2<double>));
        if (!rcd) rcd = P->ywhat1.upload(d1.data(), d1.size() * sizeof(C2<double>));
        return rcd;
    }
    std::vector<cf> h0((size_t)nent), h1((size_t)nent);
    for (int k = 0; k < nent; ++k) {
        h0[(size_t)k].re = k <= nyh ? (float)r0[(size_t)k] : 0.f; h0[(size_t)k].im = k <= nyh ? (float)i0[(size_t)k] : 0.f;
        h1[(size_t)k].re = k <= nyh ? (float)r1[(size_t)k] : 0.f; h1[(size_t)k].im = k <= nyh ? (float)i1[(size_t)k] : 0.f;
    }
    int rc = P->ywhat0.upload(h0.data(), h0.size() * sizeof(cf));
    if (!rc) rc = P->ywhat1.upload(h1.data(), h1.size() * sizeof(cf));

    return rc;
}

// The bins a unit of rows (one workgroup of pass 2) reaches, and the units that reach a bin, of a RADIAL bin map (along a half row the
// bin never decreases: row ky holds the bins r[0] .. r[nx/2]): the unit gathers, writes and has reduced the union over its rows only.
int build_unit_windows(xrfthip_plan* P, const int32_t* bm, int rpu) {
    const int nx = (int)P->ynx, nyh = (int)P->yny / 2, units = P->y_nrow_pad / rpu;
    int rcf = XRFTHIP_OK;
    std::vector<uint32_t> w((size_t)units);
    for (int un = 0; un < units; ++un) {
        int lo = P->nbins, hi = 0;
        for (int ky = un * rpu; ky < (un + 1) * rpu && ky <= nyh; ++ky) {
            lo = std::min<int>(lo, bm[(size_t)ky * nx]);
            hi = std::max<int>(hi, bm[(size_t)ky * nx + nx / 2] + 1);
        }
        if (lo > hi) lo = hi = 0;  // (a unit of padding rows only)
        w[(size_t)un] = (uint32_t)lo | (uint32_t)hi << 16;
    }
    // ... and the units that reach a bin: a contiguous range when the windows move monotonically with ky (a radial map's do;
    // otherwise every unit keeps all bins)
    bool mono = units < 65535;
    for (int un = 1; un < units && mono; ++un) {
        if ((w[(size_t)un] >> 16) == 0) continue;  // (padding rows only)
        mono = (w[(size_t)un] & 0xffffu) >= (w[(size_t)un - 1] & 0xffffu) && (w[(size_t)un] >> 16) >= (w[(size_t)un - 1] >> 16);
    }
    if (!mono) std::fill(w.begin(), w.end(), (uint32_t)P->nbins << 16);
    std::vector<uint32_t> tu((size_t)P->nbins, 0u);
    for (int b = 0; b < P->nbins; ++b) {
        int ulo = units, uhi = 0;
        for (int un = 0; un < units; ++un)
            if ((int)(w[(size_t)un] & 0xffffu) <= b && b < (int)(w[(size_t)un] >> 16)) { ulo = std::min(ulo, un); uhi = std::max(uhi, un + 1); }
        if (ulo > uhi) ulo = uhi = 0;
        tu[(size_t)b] = (uint32_t)ulo | (uint32_t)uhi << 16;
    }
    rcf = P->ytwin.upload(w.data(), w.size() * sizeof(uint32_t));
    if (!rcf) rcf = P->ytunits.upload(tu.data(), tu.size() * sizeof(uint32_t));
    return rcf;
}

// fastm: is the bin map a radial one (see fastm_rows_kernel, ISO)?  If so: first[ky][b] = the smallest |kx| <= nx/2 whose bin is >= b
// (nx/2 + 1 if none), b = 0 .. nbins, and the unit windows.  Any nx (the lengths of the table are even; odd ones would work).
int fastm_build_tfirst(xrfthip_plan* P, const int32_t* bm) {
    const int ny = (int)P->yny, nx = (int)P->ynx, nyh = ny / 2, H = nx / 2, HM = (nx - 1) / 2;
    bool radial = env_ll("XRFTHIP_ISO_GATHER", 1) != 0 && P->nbins < 65535 && H + 1 < 65535;
    for (int ky = 0; ky <= nyh && radial; ++ky) {
        const int32_t* r = bm + (size_t)ky * nx;
        const bool twin = ky != 0 && 2 * ky != ny;
        const int32_t* t = bm + (size_t)(twin ? ny - ky : ky) * nx;
        for (int m = 0; m <= H; ++m) {
            const int32_t c = r[m];
            if (c < 0 || c >= P->nbins || (m > 0 && c < r[m - 1]) || (m >= 1 && m <= HM && r[nx - m] != c)) { radial = false; break; }
            if (twin && (t[m] != c || t[(nx - m) % nx] != c)) { radial = false; break; }
        }
    }
    P->ytfirst_on = radial;
    if (!radial) return XRFTHIP_OK;
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
    int rc = P->ytfirst.upload(f.data(), f.size() * sizeof(uint16_t));
    if (!rc) rc = build_unit_windows(P, bm, fastm_gather_rpu(P));
    return rc;
}

// the bin map as pass 2 reads it (fasty_rows_kernel).  Full form: [ky < nrow_pad][kx] in natural order,
// value = (bin of (ky, kx) + 1) | (bin of the mirror (-ky, -kx) + 1) << 16; rows beyond ny/2 and unbinned samples are 0.
// Compact form, when the map has the structure of a radial one (every sample of rows 0 .. ny/2 binned; along a half row the bin
// never decreases / never increases and moves by at most one per sample; the mirror sample is in the same bin except on the
// self-mirrored rows 0 and ny/2): [ky][kx / 16] = (first sample's bin + 1) | step mask << 16 -- 1/16 of the bytes.
int fasty_build_tcodes(xrfthip_plan* P, const int32_t* bm) {
    const int ny = (int)P->yny, nx = (int)P->ynx, nyh = ny / 2;
    bool compact = env_ll("XRFTHIP_ISO_COMPACT", 1) != 0 && nx % 32 == 0;
    for (int ky = 0; ky <= nyh && compact; ++ky)
        for (int kx = 0; kx < nx; ++kx) {
            const int32_t cd = bm[(size_t)ky * nx + kx];
            if (cd < 0 || cd > 65533) { compact = false; break; }
            if (ky != 0 && ky != nyh && bm[(size_t)(ny - ky) * nx + ((nx - kx) & (nx - 1))] != cd) { compact = false; break; }
            if (kx & 15) {  // inside a segment: a step of 0 or one bin in the half row's direction
                const int32_t step = cd - bm[(size_t)ky * nx + kx - 1];
                if (step != 0 && step != (kx < nx / 2 ? 1 : -1)) { compact = false; break; }
            }
        }
    P->ytcodes_compact = compact;
    // ... and the form the atomic-free gather needs (fasty_rows_kernel): along a row the bin depends on |kx| only and never
    // decreases with it (a radial map), every sample is binned, a sample's Hermitian twin shares its bin: first[ky][b] = the
    // smallest |kx| <= nx/2 whose bin is >= b (nx/2 + 1 if none), b = 0 .. nbins
    bool radial = env_ll("XRFTHIP_ISO_GATHER", 1) != 0 && nx % 32 == 0 && P->nbins < 65535 && nx / 2 + 1 < 65535;
    for (int ky = 0; ky <= nyh && radial; ++ky) {
        const int32_t* r = bm + (size_t)ky * nx;
        for (int m = 0; m <= nx / 2; ++m) {
            const int32_t c = r[m];
            if (c < 0 || c >= P->nbins || (m > 0 && c < r[m - 1]) || (m > 0 && m < nx / 2 && r[nx - m] != c)) { radial = false; break; }
            if (ky != 0 && ky != nyh && (bm[(size_t)(ny - ky) * nx + ((nx - m) & (nx - 1))] != c || bm[(size_t)(ny - ky) * nx + m] != c)) { radial = false; break; }
        }
    }
    P->ytfirst_on = radial;
    if (radial) {
        std::vector<uint16_t> f((size_t)P->y_nrow_pad * (P->nbins + 1), (uint16_t)(nx / 2 + 1));
        for (int ky = 0; ky <= nyh; ++ky) {
            const int32_t* r = bm + (size_t)ky * nx;
            uint16_t* dst = f.data() + (size_t)ky * (P->nbins + 1);
            int m = 0;
            for (int b = 0; b <= P->nbins; ++b) {
                while (m <= nx / 2 && r[m] < b) ++m;
                dst[b] = (uint16_t)m;
            }
        }
        int rcf = P->ytfirst.upload(f.data(), f.size() * sizeof(uint16_t));
        if (rcf) return rcf;
        const bool two = P->d.out_mode == XRFTHIP_OUT_CROSS;
        const YGeomRt R = yrows_geom(P->ynx, false);
        rcf = build_unit_windows(P, bm, two ? R.gxy : R.rk);  // (rows per unit as fasty_launch_rows)
        if (rcf) return rcf;
        // the step masks of the 16-sample segments (any step size: the gather needs the run ends only)
        std::vector<uint32_t> t((size_t)P->y_nrow_pad * (nx / 16), 0u);
        for (int ky = 0; ky <= nyh; ++ky)
            for (int s0 = 0; s0 < nx; s0 += 16) {
                uint32_t w = (uint32_t)(std::min<int32_t>(bm[(size_t)ky * nx + s0], 65533) + 1);
                for (int i = 1; i < 16; ++i)
                    if (bm[(size_t)ky * nx + s0 + i] != bm[(size_t)ky * nx + s0 + i - 1]) w |= 1u << (16 + i);
                t[(size_t)ky * (nx / 16) + s0 / 16] = w;
            }
        P->ytcodes_compact = true;  // (the table has the compact form; the gather reads its masks only)
        return P->ytcodes.upload(t.data(), t.size() * sizeof(uint32_t));
    }
    if (compact) {
        std::vector<uint32_t> t((size_t)P->y_nrow_pad * (nx / 16), 0u);
        for (int ky = 0; ky <= nyh; ++ky)
            for (int s0 = 0; s0 < nx; s0 += 16) {
                uint32_t w = (uint32_t)(bm[(size_t)ky * nx + s0] + 1);
                for (int i = 1; i < 16; ++i)
                    if (bm[(size_t)ky * nx + s0 + i] != bm[(size_t)ky * nx + s0 + i - 1]) w |= 1u << (16 + i);
                t[(size_t)ky * (nx / 16) + s0 / 16] = w;
            }
        return P->ytcodes.upload(t.data(), t.size() * sizeof(uint32_t));
    }
    std::vector<uint32_t> t((size_t)P->y_nrow_pad * nx, 0u);
    for (int ky = 0; ky <= nyh; ++ky)
        for (int kx = 0; kx < nx; ++kx) {
            uint32_t v = 0;
            const int32_t cd = bm[(size_t)ky * nx + kx];
            if (cd >= 0) v |= (uint32_t)(cd + 1);
            if (ky != 0 && ky != nyh) {
                const int32_t cm = bm[(size_t)(ny - ky) * nx + ((nx - kx) & (nx - 1))];
                if (cm >= 0) v |= (uint32_t)(cm + 1) << 16;
            }
            t[(size_t)ky * nx + kx] = v;
        }
    return P->ytcodes.upload(t.data(), t.size() * sizeof(uint32_t));
}
// the radial-sum tables of one round share the transforms' LDS with the staged half of the workgroup's rows: they must fit the other half
bool fasty_iso_tables_fit(const xrfthip_plan* P, int nbins) {
    const YGeomRt R = yrows_geom(P->ynx);
    const size_t half = (size_t)R.gxy * (size_t)(P->ynx + P->ynx / 16) * 4;  // GX rows of floats = GX / 2 rows of complex
    const size_t hw = P->d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1;
    return nbins <= 65534 && (size_t)nbins * (8 * hw + 4) <= half;
}

bool fasty_on(const xrfthip_plan* P) { return P->yfirst && fast_on(P); }

// Workgroups of `kernel` the whole device holds at once (a persistent launch's grid): the occupancy calculator's count per CU times the CUs,
// asked once per kernel.
long long resident_workgroups(const void* kernel, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<const void*, long long> memo;
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(kernel);
    if (it != memo.end()) return it->second;
    int per_cu = 0, cus = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const long long n = (long long)per_cu * cus;
    memo[kernel] = n;
    return n;
}

void fasty_launch_cols(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof) {
    const xrfthip_desc& d = P->d;
    const YGeomRt C = ycols_geom(P->yny);
    xrfthip_plan::ProfRec* rec = prof ? prof_begin(P, "fasty_cols", st) : nullptr;
    const dim3 grid((unsigned)(gc * (P->ynx / C.cw))), blk((unsigned)C.thr);
#define YC_(NN) do { if (d.detrend) { auto k = &fasty_cols_kernel<NN, true>; XRFT_LAUNCH(k, grid, blk, C.lds, st, p); } \
                     else { auto k = &fasty_cols_kernel<NN, false>; XRFT_LAUNCH(k, grid, blk, C.lds, st, p); } } while (0)
#define YCW_(NN) do { if (d.detrend) { auto k = &fasty_cols_kernel<NN, true, true>; XRFT_LAUNCH(k, grid, blk, C.lds, st, p); } \
                      else { auto k = &fasty_cols_kernel<NN, false, true>; XRFT_LAUNCH(k, grid, blk, C.lds, st, p); } } while (0)
    if (P->fast1d_win) {  // four-step 1-D with a window: the slab-shaped window table
        if (P->yny == 4096) YCW_(4096); else if (P->yny == 2048) YCW_(2048); else if (P->yny == 1024) YCW_(1024); else if (P->yny == 512) YCW_(512); else YCW_(256);
    }
    else if (P->yny == 4096) YC_(4096); else if (P->yny == 2048) YC_(2048); else if (P->yny == 1024) YC_(1024); else if (P->yny == 512) YC_(512); else YC_(256);
#undef YC_
#undef YCW_
    prof_end(rec, st);
    if (d.detrend) {  // plane (2-D) or line through the whole sequence (four-step 1-D) from the per-column sums -> what pass 2 has to add back
        rec = prof ? prof_begin(P, "fasty_fit", st) : nullptr;
        if (P->fast1d) { auto kf = &fasty_fit1d_kernel; XRFT_LAUNCH(kf, dim3((unsigned)gc), dim3(256), 3 * 256 * sizeof(double), st, (const double*)p.colfit, const_cast<float*>(p.corr), (int)P->ynx, (int)P->yny, (int)d.detrend); }
        else { auto kf = &fasty_fit_kernel; XRFT_LAUNCH(kf, dim3((unsigned)gc), dim3(256), 3 * 256 * sizeof(double), st, (const double*)p.colfit, p.win_x, const_cast<float*>(p.corr), (int)P->ynx, (int)P->yny, (int)d.detrend); }
        prof_end(rec, st);
    }
}

void fasty_launch_rows(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof) {
    const xrfthip_desc& d = P->d;
    const YGeomRt R = yrows_geom(P->ynx, P->fast1d);
    const bool iso_on = (d.flags & XRFTHIP_ISO) != 0;
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    xrfthip_plan::ProfRec* rec = prof ? prof_begin(P, "fasty_rows", st) : nullptr;
    const int rpu = two ? R.gxy : R.rk;  // a cross spectrum spends both transforms of a thread on one row (field 0, field 1)
    // (four-step: rows 0 .. ny/2 - 1 in whole units, the Nyquist rows of R.gxy consecutive slabs in one extra unit each)
    const dim3 grid((unsigned)(P->fast1d ? gc * ((P->yny / 2) / rpu) + (gc + R.gxy - 1) / R.gxy : gc * (P->y_nrow_pad / rpu))), blk((unsigned)R.thr);
    const int hw = d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1;
    const size_t lds = R.lds;  // (the radial-sum tables alias the transforms' LDS)
    // nothing but the radial sums of a radial map leaves the pass: the persistent kernel of fasty_iso.h (as many workgroups as the chip holds)
    // (measured, profiles/r06_tune_iso.txt: with the pipelined gather in BOTH kernels the workgroup-per-unit kernel is level or ahead -- 19.1 against 19.5 us per 4096^2
    // slab, 4.87 against 5.26 at 2048^2, 1.21 against 1.20 at 1024^2 -- so the persistent kernel is opt-in: XRFTHIP_ISOROWS=1; =2 its profiling build)
    const bool iso_persistent = P->tune_isorows == 2 || P->tune_isorows == 1;
    if (iso_on && p.out == nullptr && p.tfirst != nullptr && !P->fast1d && iso_persistent && P->ynx >= 1024 &&
        (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_CROSS)) {
        const long long total = gc * (P->y_nrow_pad / rpu);
        const bool tim = P->tune_isorows == 2;
        FastY q = p;
        void (*kern)(FastY) = nullptr;
#define YI_(NN) kern = d.out_mode == XRFTHIP_OUT_POWER ? (tim ? &fasty_isorows_kernel<NN, 1, true> : &fasty_isorows_kernel<NN, 1, false>) \
                                                       : (tim ? &fasty_isorows_kernel<NN, 2, true> : &fasty_isorows_kernel<NN, 2, false>)
        if (P->ynx == 4096) YI_(4096); else if (P->ynx == 2048) YI_(2048); else YI_(1024);
#undef YI_
        const long long slots = resident_workgroups(reinterpret_cast<const void*>(kern), R.thr, lds);
        const unsigned nblk = (unsigned)std::min<long long>(total, slots);
        if (tim) {
            std::vector<long long> z((size_t)nblk * 8, 0);
            if (P->iso_tim.upload(z.data(), z.size() * sizeof(long long)) == XRFTHIP_OK) q.tim = reinterpret_cast<long long*>(P->iso_tim.p);
        }
        XRFT_LAUNCH(kern, dim3(nblk), blk, lds, st, q);
        prof_end(rec, st);
        if (tim && q.tim) {
            (void)hipStreamSynchronize(st);
            std::vector<long long> h((size_t)nblk * 8, 0);
            (void)hipMemcpy(h.data(), q.tim, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (unsigned b = 0; b < nblk; ++b) for (int i = 0; i < 8; ++i) acc[i] += (double)h[(size_t)b * 8 + i];
            const double units = std::max(acc[7], 1.0);
            std::fprintf(stderr, "[xrfthip isorows nx=%lld mode=%d] %u workgroups, %.0f units; shader-clock cycles per unit: wait+tables+addback %.0f | fft %.0f | stage+barrier %.0f | "
                         "prefetch issue %.0f | segments %.0f | barrier %.0f | gather %.0f | total %.0f\n", (long long)P->ynx, (int)d.out_mode, nblk, units, acc[0] / units, acc[1] / units,
                         acc[2] / units, acc[3] / units, acc[4] / units, acc[5] / units, acc[6] / units, (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6]) / units);
        }
    } else {
#define YR_(NN) do { \
        if (d.out_mode == XRFTHIP_OUT_POWER) { if (iso_on) { auto k = &fasty_rows_kernel<NN, 1, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } else { auto k = &fasty_rows_kernel<NN, 1, false>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } } \
        else if (d.out_mode == XRFTHIP_OUT_CROSS) { if (iso_on) { auto k = &fasty_rows_kernel<NN, 2, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } else { auto k = &fasty_rows_kernel<NN, 2, false>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } } \
        else if (d.out_mode == XRFTHIP_OUT_PHASE) { auto k = &fasty_rows_kernel<NN, 3, false>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } \
        else { auto k = &fasty_rows_kernel<NN, 0, false>; XRFT_LAUNCH(k, grid, blk, lds, st, p); } } while (0)
    if (P->fast1d) {  // four-step 1-D: rows of 256 samples, transposed stores
        if (P->fast1d_win) {
            if (d.out_mode == XRFTHIP_OUT_POWER) { auto k = &fasty_rows_kernel<256, 1, false, true, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); }
            else { auto k = &fasty_rows_kernel<256, 0, false, true, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); }
        }
        else if (d.out_mode == XRFTHIP_OUT_POWER) { auto k = &fasty_rows_kernel<256, 1, false, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); }
        else { auto k = &fasty_rows_kernel<256, 0, false, true>; XRFT_LAUNCH(k, grid, blk, lds, st, p); }
    }
    else if (P->ynx == 4096) YR_(4096); else if (P->ynx == 2048) YR_(2048); else if (P->ynx == 1024) YR_(1024); else if (P->ynx == 512) YR_(512); else YR_(256);
#undef YR_
    prof_end(rec, st);
    }
    if (iso_on) {  // the row workgroups' partial sums, added in order
        rec = prof ? prof_begin(P, "fasty_iso_reduce", st) : nullptr;
        const int nb = P->nbins * hw, upr = P->y_nrow_pad / rpu;
        auto kr = &iso_reduce_kernel;
        XRFT_LAUNCH(kr, dim3((unsigned)((nb + 63) / 64), (unsigned)gc), dim3(256), 4 * 64 * sizeof(double), st, (const double*)p.iso_part, p.iso, upr, nb, P->ytfirst_on ? reinterpret_cast<const unsigned*>(P->ytunits.p) : nullptr, hw);
        prof_end(rec, st);
    }
}

// parameter block of one group of slabs [g0, g0 + gc): the intermediate and the fit tables sit in ring slot `slot` (of slot_slabs slabs each)
FastY fasty_params(const xrfthip_plan* P, const float* in, void* out, double* iso, char* ws, long long g0, long long gc, int slot, long long slot_slabs) {
    // (slot 0 = field 0 / the only field, slot 1 = field 1 of a cross spectrum: its own intermediate and fit tables)
    const xrfthip_desc& d = P->d;
    const size_t slab_pts = (size_t)P->yny * P->ynx;
    const bool want_out = !(d.flags & XRFTHIP_NO_SPECTRUM_OUT);
    const bool iso_on = (d.flags & XRFTHIP_ISO) != 0;
    const YGeomRt C = ycols_geom(P->yny);
    const size_t s0 = (size_t)slot * slot_slabs;  // first slab of the slot inside the workspace arrays
    FastY p{};
    p.in = in + (size_t)g0 * slab_pts;
    p.w2 = reinterpret_cast<cf*>(ws + P->off_w) + s0 * (size_t)P->y_nrow_pad * P->ynx;
    const size_t out_esz = (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_PHASE) ? sizeof(float) : sizeof(cf);
    const size_t out_pts = (size_t)P->yny * ((d.flags & XRFTHIP_HALF_X) ? P->ynx / 2 + 1 : P->ynx);
    p.out = want_out ? (char*)out + (size_t)g0 * out_pts * out_esz : nullptr;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0;
    p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.ph_y = reinterpret_cast<const cf*>(P->fph[0].p);
    p.ph_x = reinterpret_cast<const cf*>(P->fph[1].p);
    p.tw_big = reinterpret_cast<const cf*>(P->tw_big1d.p);
    p.ph_on = P->fph_on ? 1 : 0;
    p.tw_x = reinterpret_cast<const cf*>(P->tw_fx.p);
    p.tw_y = reinterpret_cast<const cf*>(P->tw_fy.p);
    p.win_y = reinterpret_cast<const float*>(P->win[0].p ? P->win[0].p : P->ones4096.p);
    p.win_x = reinterpret_cast<const float*>(P->win[1].p ? P->win[1].p : P->ones4096.p);
    p.colfit = reinterpret_cast<double*>(ws + P->off_rowfit) + s0 * (size_t)P->ynx * 4;
    p.corr = reinterpret_cast<const float*>(ws + P->off_corr) + s0 * (size_t)P->ynx * 2;
    p.what0 = reinterpret_cast<const cf*>(P->ywhat0.p);
    p.what1 = reinterpret_cast<const cf*>(P->ywhat1.p);
    p.tcodes = reinterpret_cast<const unsigned*>(P->ytcodes.p);
    p.tcodes_compact = P->ytcodes_compact ? 1 : 0;
    p.tfirst = P->ytfirst_on ? reinterpret_cast<const unsigned short*>(P->ytfirst.p) : nullptr;
    p.twin = P->ytfirst_on ? reinterpret_cast<const unsigned*>(P->ytwin.p) : nullptr;
    p.iso = iso_on ? iso + (size_t)g0 * P->nbins * (d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1) : nullptr;
    p.nbins = P->nbins;
    p.iso_part = reinterpret_cast<double*>(ws + P->off_isopart);
    p.ny = (int)P->yny; p.nx = (int)P->ynx;
    p.nrow_pad = P->y_nrow_pad;
    p.l_cw = ilog2i(C.cw); p.l_rk = ilog2i(C.rk); p.l_2gy = ilog2i(2 * C.gxy);
    p.detrend = d.detrend;
    p.nslab = (int)gc;
    p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(P->yny / 2) : 0;
    p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(P->ynx / 2) : 0;  // (four-step 1-D: the shift by N/2 samples is k2 + nx/2)
    if (P->fast1d) p.win_y = p.win_x = reinterpret_cast<const float*>(P->ones4096.p);  // (a window of the whole sequence: win2d)
    p.win2d = reinterpret_cast<const float*>(P->fast1d_win ? P->win2d.p : nullptr);
    p.scale = (float)d.scale;
    p.tune = (int)P->tune_y;
    return p;
}

int run_fasty(const xrfthip_plan* P, const float* in, const float* in1, void* out, double* iso, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        FastY p = fasty_params(P, in, out, iso, ws, g0, gc, 0, P->G);
        if ((P->tune_y >> 21) & 1) {
            p.rdv = reinterpret_cast<unsigned*>(ws + P->off_rdv);
            HIP_TRY(hipMemsetAsync(p.rdv, 0, (size_t)gc * (size_t)std::max<long long>(P->ynx / 8, 1) * sizeof(unsigned), st));
        }
        fasty_launch_cols(P, p, gc, st, true);
        if (two) {  // field 1 through the same column pass into its own intermediate; the row pass reads both
            const FastY p1 = fasty_params(P, in1, out, iso, ws, g0, gc, 1, P->G);
            fasty_launch_cols(P, p1, gc, st, true);
            p.w2b = p1.w2;
            p.corr_b = p1.corr;
        }
        fasty_launch_rows(P, p, gc, st, true);
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

// the two-pass pipeline on complex float32 slabs (fasty_c2c.h): columns -> rows, group by group
int run_fastyc(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st) {
    const xrfthip_desc& d0 = P->d;
    // (the four-step form: every batch entry is ONE sequence of d.nx points, transformed as the [d.nx / 256][256] view)
    struct { long long batch, ny, nx; uint32_t flags; int out_mode; double scale; } d{d0.batch, P->fastyc_fs ? d0.nx / 256 : d0.ny, P->fastyc_fs ? 256 : d0.nx, d0.flags, d0.out_mode, d0.scale};
    const bool fs = P->fastyc_fs;
    const bool c2r = (d.flags & XRFTHIP_C2R_X) != 0;
    const long long nxt = c2r ? d.nx / 2 : d.nx;  // points of the row transforms
    const YGeomRt C = ycols_geom(d.ny), R = yrows_geom(nxt);
    const int cw = 2 * C.gxy, rk = std::max(1, 16 / cw);
    const int nxb_full = (int)(nxt / cw), w2_nxb = nxb_full + (c2r ? 1 : 0);
    const long long in_pitch = c2r ? d.nx / 2 + 1 : d.nx;
    const size_t lds_c = (size_t)(C.gxy * (ycols_gstr(d.ny)) + 16 * (d.ny / 256)) * sizeof(cf);
    const size_t out_esz = (d.out_mode == XRFTHIP_OUT_POWER || c2r) ? sizeof(float) : sizeof(cf);
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        FastYC p{};
        p.in = reinterpret_cast<const cf*>(in) + (size_t)g0 * d.ny * in_pitch;
        p.c2r = c2r ? 1 : 0; p.in_pitch = (int)in_pitch; p.w2_nxb = w2_nxb;
        p.tw_big = reinterpret_cast<const cf*>(P->tw_big1d.p);
        p.w2 = reinterpret_cast<cf*>(ws + P->off_w);
        p.out = (char*)out + (size_t)g0 * d.ny * d.nx * out_esz;
        p.tw_x = reinterpret_cast<const cf*>(P->tw_fx.p);
        p.tw_y = reinterpret_cast<const cf*>(P->tw_fy.p);
        p.win_y = reinterpret_cast<const float*>(P->win[0].p ? P->win[0].p : P->ones4096.p);
        p.win_x = reinterpret_cast<const float*>(P->win[1].p ? P->win[1].p : P->ones4096.p);
        p.win_on = (P->win[0].p || P->win[1].p) ? 1 : 0;
        p.ph_y = reinterpret_cast<const cf*>(P->fph[0].p);
        p.ph_x = reinterpret_cast<const cf*>(P->fph[1].p);
        const bool phase = d.out_mode == XRFTHIP_OUT_COMPLEX && P->fph_on;
        p.ph_in = (phase && (d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
        p.ph_on = (phase && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
        p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
        p.ishift_y = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_Y)) ? 1 : 0;  // (a forward plan's ifftshifted input is the sign (-1)^k in the phase tables)
        p.ishift_x = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_X)) ? 1 : 0;
        p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
        p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
        if (fs) {
            p.fs = 1;
            // the sequence rotated by N / 2 = the view's rows rotated by ny / 2; its fftshift = the row transforms' output rotated by 128; an input phase in its
            // separable form (finalize_plan), an output phase as the table over the whole sequence
            p.ishift_y = p.ishift_x; p.ishift_x = 0; p.shift_y = 0;
            if (p.ph_in) p.ph_x = reinterpret_cast<const cf*>(P->fs_phx.p);
        }
        p.ny = (int)d.ny; p.nx = (int)d.nx; p.nslab = (int)gc;
        p.l_cw = ilog2i(cw); p.l_rk = ilog2i(rk);
        p.power = d.out_mode == XRFTHIP_OUT_POWER ? 1 : 0;
        p.scale = (float)d.scale;
        xrfthip_plan::ProfRec* rec = prof_begin(P, "fastyc_cols", st);
        const dim3 gridc((unsigned)(gc * nxb_full + (c2r ? gc : 0))), blkc((unsigned)C.thr);
#define YCC_(NN) do { auto k = &fastyc_cols_kernel<NN>; XRFT_LAUNCH(k, gridc, blkc, lds_c, st, p); } while (0)
        if (d.ny == 4096) YCC_(4096); else if (d.ny == 2048) YCC_(2048); else if (d.ny == 1024) YCC_(1024); else if (d.ny == 512) YCC_(512); else YCC_(256);
#undef YCC_
        prof_end(rec, st);
        rec = prof_begin(P, "fastyc_rows", st);
        const dim3 gridr((unsigned)(gc * (d.ny / R.rk))), blkr((unsigned)R.thr);
#define YCR_(NN) do { auto k = &fastyc_rows_kernel<NN>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
#define YC2_(NN) do { auto k = &fastyc_rows_c2r_kernel<NN, false>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
        if (c2r) { if (nxt == 2048) YC2_(2048); else if (nxt == 1024) YC2_(1024); else if (nxt == 512) YC2_(512); else YC2_(256); }
        else if (d.nx == 4096) YCR_(4096); else if (d.nx == 2048) YCR_(2048); else if (d.nx == 1024) YCR_(1024); else if (d.nx == 512) YCR_(512); else YCR_(256);
#undef YCR_
#undef YC2_
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}


// kernels of this unit that take more than 64 KB of dynamic LDS (the y-first float32 kernels): called once through set_kernel_attrs_once()
void set_attrs_fasty() {
    const int m = (int)kLdsMax;
#define SETF(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, m)
#define SETY(NN) SETF((fasty_cols_kernel<NN, false>)); SETF((fasty_cols_kernel<NN, true>)); SETF((fasty_cols_kernel<NN, false, true>)); SETF((fasty_cols_kernel<NN, true, true>)); SETF((fasty_rows_kernel<NN, 1, false>)); SETF((fasty_rows_kernel<NN, 1, true>)); \
                 SETF((fasty_rows_kernel<NN, 0, false>)); SETF((fasty_rows_kernel<NN, 2, false>)); SETF((fasty_rows_kernel<NN, 2, true>)); SETF((fasty_rows_kernel<NN, 3, false>))
    SETY(4096); SETY(2048); SETY(1024); SETY(512); SETY(256);
#undef SETY
#define SETC(NN) SETF((fastyc_cols_kernel<NN>)); SETF((fastyc_rows_kernel<NN>))
    SETC(4096); SETC(2048); SETC(1024); SETC(512); SETC(256);
    SETF((fastyc_rows_c2r_kernel<2048, false>)); SETF((fastyc_rows_c2r_kernel<1024, false>)); SETF((fastyc_rows_c2r_kernel<512, false>)); SETF((fastyc_rows_c2r_kernel<256, false>));
    SETF((fastyc_rows_c2r_kernel<2048, true>)); SETF((fastyc_rows_c2r_kernel<1024, true>)); SETF((fastyc_rows_c2r_kernel<512, true>)); SETF((fastyc_rows_c2r_kernel<256, true>));
#undef SETC
#define SETI(NN) SETF((fasty_isorows_kernel<NN, 1, false>)); SETF((fasty_isorows_kernel<NN, 2, false>)); SETF((fasty_isorows_kernel<NN, 1, true>)); SETF((fasty_isorows_kernel<NN, 2, true>))
    SETI(4096); SETI(2048); SETI(1024);
#undef SETI
    SETF((fasty_rows_kernel<256, 0, false, true>)); SETF((fasty_rows_kernel<256, 1, false, true>));
    SETF((fasty_rows_kernel<256, 0, false, true, true>)); SETF((fasty_rows_kernel<256, 1, false, true, true>));
#undef SETF
}

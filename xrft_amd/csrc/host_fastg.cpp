#include "plan.h"

// ---------------------------------------------------------------------------------------------------------------
// one pass over small slabs of any smooth shape (fastg.h): the lengths as data
// ---------------------------------------------------------------------------------------------------------------
int fastg_rev(const std::vector<int>& radix, int n, DevBuf& buf, std::vector<unsigned>& host) {  // rev[k] = position of frequency k after the DIF passes (as build_tables)
    std::vector<unsigned> rev((size_t)std::max(n, 1));
    for (int pos = 0; pos < n; ++pos) {
        if (radix.empty()) { rev[(size_t)pos] = (unsigned)pos; continue; }  // (no passes: the identity)
        long long L = n, rem = pos, k = 0, mult = 1;
        for (int r : radix) {
            const long long m = L / r;
            k += (rem / m) * mult;
            rem %= m;
            mult *= r;
            L = m;
        }
        rev[(size_t)k] = (unsigned)pos;
    }
    host = rev;
    return buf.upload(rev.data(), rev.size() * sizeof(unsigned));
}
template <typename T> int fastg_setup_t(xrfthip_plan* P) {
    const xrfthip_desc& d = P->d;
    const int n = P->g_n, ny = P->g_one_d ? P->g_rows : (int)d.ny;
    int rc = build_twiddle<T>(P->g_twx, n, n);
    if (!rc) rc = build_twiddle<T>(P->g_twy, ny, ny);
    if (!rc && P->g_packed) rc = build_twiddle<T>(P->g_twr, d.nx, n + 1);
    if (!rc) rc = fastg_rev(P->g_rx, n, P->g_revx, P->g_hrevx);
    if (!rc) rc = fastg_rev(P->g_ry, ny, P->g_revy, P->g_hrevy);  // (no passes: the identity)
    return rc;
}
// the radices of one axis of the lengths-as-data one-pass kernels (fastg.h): the 2^a 3^b 5^c choice of `factorize` where the length is that smooth (unchanged
// plans), else -- prime factors 7, 11, 13: weekly data, 77, 91, 364 = 52 weeks, 1001 -- the butterflies of tile_fft.h's dft_prime (numpy's pocketfft
// hard-codes 7 and 11).  False: another prime factor.
bool fastg_factor(long long n, std::vector<int>& out) {
    bool gen = false;
    if (factorize(n, out, gen) == XRFTHIP_OK && !gen) return true;
    if (n == 7 || n == 11 || n == 13 || n == 14) { out.assign(1, (int)n); return true; }
    return fastn_factor(n, 16, out);
}
bool fastg_try(xrfthip_plan* P) {  // can the slab's half spectrum live in the LDS of one workgroup, and are both lengths smooth?
    const xrfthip_desc& d = P->d;
    const bool one_d = d.ndim == 1;
    if ((d.ndim != 2 && !one_d) || d.nx < 3 || (!one_d && d.ny < 2) || d.nx > (one_d ? 16384 : 4096) || d.ny > 4096) return false;
    // an even nx: the rows packed in pairs of samples, the half spectrum (nx / 2 + 1 columns) in the tile; an odd nx: the rows as complex sequences with
    // zero imaginary parts, the whole spectrum in the tile (twice the LDS and the x passes' work: 75 x 75, 81 x 81, 125 x 125 boxes)
    const bool c2r = (d.flags & XRFTHIP_C2R_X) != 0;  // (irfftn: the half spectrum in, the packed geometry)
    const bool packed = !(d.nx & 1) && (!P->cplx_in || c2r);  // (complex input: every row a complex sequence, the whole spectrum in the tile)
    const int n = packed ? (int)(d.nx / 2) : (int)d.nx;
    int rs = packed ? n + 1 : n;
    if (!(rs & 1)) ++rs;  // an odd row stride: the rows' passes and the gather of the output loop spread over the banks
    // a 1-D transform along x: groups of rows as "slabs" without y passes -- as many rows as make a tile of ~24 KB (several workgroups per CU), 1 ... 256
    int ny = (int)d.ny;
    if (one_d) {
        ny = (int)std::max<long long>(1, std::min<long long>(256, (24 * 1024) / ((long long)rs * (long long)P->csize)));
        if (ny < 2 && !P->cplx_in) return false;  // (one long real row per workgroup: the row tiles of the generic passes do as well -- (8192, 3000) float64 65 vs 46 GFFT/s;
                                                  //  complex rows -- inverse transforms -- run 40 GFFT/s there: taken)
        ny = (int)std::min<long long>(ny, std::max<long long>(1, d.batch));
    }
    const size_t nf = d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1;  // a cross spectrum holds both fields' tiles
    const int nred = one_d ? std::max(2 * ny, kFastGWaves * 3) : kFastGWaves * 3;
    const size_t lds = (((size_t)nf * ny * rs * P->csize + 15) & ~(size_t)15) + (size_t)(n + ny + n + 1) * P->csize + (size_t)nred * sizeof(double) +
                       (size_t)(ny + d.nx) * P->rsize + (((size_t)n * 2 + 3) & ~(size_t)3) + (size_t)ny * 2 + 16;  // the tile + the plan's tables, the windows + the wave sums
    if (lds > kLdsMax - 1024) return false;
    bool gx = false, gy = false;
    std::vector<int> rx, ry;
    (void)gx; (void)gy;
    if (n == 1) rx.clear(); else if (!fastg_factor(n, rx)) return false;
    if (!one_d && !fastg_factor(ny, ry)) return false;
    if ((int)rx.size() > kFastGMaxPasses || (int)ry.size() > kFastGMaxPasses) return false;
    for (int r : rx) if (r > 16) return false;
    for (int r : ry) if (r > 16) return false;
    P->g_rx = rx; P->g_ry = ry; P->g_rs = rs; P->g_lds = lds; P->g_n = n; P->g_packed = packed;
    P->g_one_d = one_d; P->g_rows = ny; P->g_nred = nred;
    return true;
}
// fastg radial sums: per bin the LDS positions of its samples, in (ky, kx) order -- any bin map (a sample with kx > nx/2 lives at its Hermitian twin's
// position: |F|^2 is the same)
int fastg_build_iso(xrfthip_plan* P, const int32_t* bm) {
    const int ny = (int)P->d.ny, nx = (int)P->d.nx, n = P->g_n, rs = P->g_rs, nb = P->nbins;
    const bool packed = P->g_packed;
    const bool cross = P->d.out_mode == XRFTHIP_OUT_CROSS;  // (bit 15 of a position: the sample is the conjugate of the stored product)
    if ((size_t)ny * rs > (cross ? 32767u : 65535u) || nb < 1) { P->fastg = false; return XRFTHIP_OK; }  // (16-bit positions; the other paths take the plan)
    std::vector<unsigned> start((size_t)nb + 1, 0u);
    for (size_t e = 0; e < (size_t)ny * nx; ++e) if (bm[e] >= 0 && bm[e] < nb) ++start[(size_t)bm[e] + 1];
    for (int b = 0; b < nb; ++b) start[(size_t)b + 1] += start[(size_t)b];
    std::vector<unsigned> fill(start.begin(), start.end() - 1);
    std::vector<uint16_t> pos(std::max<size_t>(1, start[(size_t)nb]));
    for (int ky = 0; ky < ny; ++ky)
        for (int kx = 0; kx < nx; ++kx) {
            const int32_t c = bm[(size_t)ky * nx + kx];
            if (c < 0 || c >= nb) continue;
            const bool mir = packed && kx > n;
            const int sy = mir ? (ky == 0 ? 0 : ny - ky) : ky, sx = mir ? nx - kx : kx;
            pos[fill[(size_t)c]++] = (uint16_t)((P->g_hrevy[(size_t)sy] * (unsigned)rs + ((packed && sx == n) ? (unsigned)n : P->g_hrevx[(size_t)sx])) | ((cross && mir) ? 0x8000u : 0u));
        }
    int rc = P->g_isopos.upload(pos.data(), pos.size() * sizeof(uint16_t));
    if (!rc) rc = P->g_isostart.upload(start.data(), start.size() * sizeof(unsigned));
    return rc;
}

// one transform axis that is not the contiguous one, any smooth length (fastg.h: fastgy_kernel): G complex sequences = 2 G real columns per workgroup,
// the widest power of two (<= 128 bytes of a row) whose tile leaves three workgroups on a CU, or the widest that fits at all
// n = q p, p ONE prime 17 ... 127 whose p - 1 the butterflies factor, q smooth and prime to p: the prime-factor form with Rader's algorithm along p (fastg.h)
bool rader_split(long long n, bool allow17, int& p_out, std::vector<int>& rq, std::vector<int>& rp) {
    if (!env_ll("XRFTHIP_RADER", 1)) return false;
    for (int p = 17; p <= 127; ++p) {
        bool prime = true;
        for (int f = 2; f * f <= p; ++f) if (p % f == 0) { prime = false; break; }
        if (!prime || n % p) continue;
        const long long q = n / p;
        if (q % p == 0) return false;  // (p^2)
        rq.clear(); rp.clear();
        if (q > 1 && !fastg_factor(q, rq)) return false;  // (a second prime without a butterfly)
        if (!fastg_factor(p - 1, rp)) {
            // 103 - 1 = 6 x 17 (the ERA5 grid's 721 = 7 x 103 latitudes): the 17-point butterfly, which only the Rader forms carry -- the two-pass pipeline's columns in both
            // precisions (float64: 6 spilled registers), the one-axis kernel in float32 only (float64: 256 registers, one wave per SIMD; measured 76 -> 46 GFFT/s)
            if (!allow17 || (p - 1) % 17 || !fastg_factor((p - 1) / 17, rp)) return false;
            rp.push_back(17);
        }
        for (int r : rq) if (r > 16) return false;
        for (int r : rp) if (r > 17) return false;
        if ((int)rq.size() > kNMaxPass || (int)rp.size() > kNMaxPass) return false;
        if ((int)rq.size() > kFastGMaxPasses || (int)rp.size() > kFastGMaxPasses) return false;
        p_out = p;
        return true;
    }
    return false;
}

bool fastgy_try(xrfthip_plan* P, bool rows) {
    const xrfthip_desc& d = P->d;
    const bool two_f = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    const long long N = rows ? d.nx : d.ny;  // the transform length
    if (rows) { if (d.ndim != 1 || d.nx < 17 || d.nx > 4096 || d.batch >= (1LL << 31)) return false; }
    else if (d.ndim != 2 || (!P->cplx_in && !two_f && (d.nx & 1)) || d.ny < 2 || d.ny > 16384 || d.batch * ((d.nx + 3) / 4) >= (1LL << 31)) return false;
    bool gy = false;
    std::vector<int> ry, rp;
    long long m = N;  // rows of the tile = length of the passes: ny, or the Bluestein length when a prime factor of ny has no butterfly
    int blue_m = 0, rad_p = 0;
    if (!fastg_factor(N, ry) && N <= 4096 && rader_split(N, !P->dbl, rad_p, ry, rp)) {
        // (ry: the radices of q; the tile holds ny rows)
    } else if (rows) {
        return false;  // (the contiguous axis: smooth lengths have fastg_kernel's row groups, the others the generic passes)
    } else if (!fastg_factor(d.ny, ry)) {
        rad_p = 0;
        for (m = 2 * d.ny - 1;; ++m) {
            long long q = m;
            while (q % 2 == 0) q /= 2;
            while (q % 3 == 0) q /= 3;
            while (q % 5 == 0) q /= 5;
            if (q == 1) break;
        }
        if (m > 65535 || factorize(m, ry, gy) || gy) return false;
        blue_m = (int)m;
    }
    if ((int)ry.size() > kFastGMaxPasses) return false;
    for (int r : ry) if (r > 16) return false;
    if (d.ny > 4096 && !blue_m) return false;
    // threads by the points of the tile: a short axis on 256 threads leaves most waves idle at every barrier -- (48, 1024, 1024) float32 on one wave 236 GFFT/s against
    // 160, (96, 512, 512) on two 243 against 217, float64 126 against 98; from ~2400 points on: 256 (profiles/r05_gy_threads.txt)
    const long long thr_env = env_ll("XRFTHIP_FASTGY_THR", 0);
    auto thr_of = [&](int G) {
        if (thr_env) return (int)std::min<long long>(256, std::max<long long>(64, thr_env / 64 * 64));
        const long long pts = (long long)G * m;
        return P->dbl ? (pts <= 512 ? 64 : pts <= 1024 ? 128 : 256) : (pts <= 1024 ? 64 : pts <= 2048 ? 128 : 256);
    };
    auto lds_of = [&](int G, bool tw_lds) {
        const int thr = thr_of(G);
        return (((size_t)m * (rows ? G + 1 : G) * P->csize + 15) & ~(size_t)15) + (tw_lds ? (size_t)(rad_p ? m / rad_p : m) * P->csize : 0) + (size_t)thr * 4 * sizeof(double) + (size_t)G * 4 * sizeof(double) +
               (size_t)N * P->rsize + (size_t)N * 2 + 16 + (rad_p ? (((size_t)N * 2 + 15) & ~(size_t)15) + (size_t)rad_p * P->csize + 16 : 0);
    };
    const int gmax = (int)(128 / P->csize);  // 128 bytes of a row: 16 float32 pairs, 8 float64 pairs
    int G = 0;
    bool tw_lds = true;
    for (int cand = gmax; cand >= gmax / 4 && cand >= 1 && !G; cand >>= 1) if (lds_of(cand, true) <= 78 * 1024) G = cand;   // two or more workgroups per CU, 32 bytes of a row at least
    for (int cand = gmax; cand >= 1 && !G; cand >>= 1) if (lds_of(cand, true) <= kLdsMax - 1024) G = cand;                  // ... or whatever fits
    if (!G && blue_m) {  // (a Bluestein tile that leaves no room for the twiddles: they come from memory)
        tw_lds = false;
        for (int cand = gmax; cand >= 1 && !G; cand >>= 1) if (lds_of(cand, false) <= kLdsMax - 1024) G = cand;
    }
    const long long forced = env_ll("XRFTHIP_FASTGY_G", 0);
    if (forced >= 1 && forced <= gmax && !(forced & (forced - 1)) && lds_of((int)forced, tw_lds) <= kLdsMax - 1024) G = (int)forced;
    if (!G) return false;
    P->g_ry = ry; P->gy_G = G; P->gy_thr = thr_of(G); P->gy_lds = lds_of(G, tw_lds); P->gy_blue_m = blue_m; P->gy_tw_lds = tw_lds;
    P->gy_rad_p = rad_p; P->gy_rp = rp; P->gy_rows = rows; P->gy_n = N;
    return true;
}
// the tables of the prime-factor / Rader form (fastg.h, FastGY::rad_p): the row of every input sample and of every frequency, the transformed kernel
// B = FFT_(p-1)(W_p^(g^m)) / (p - 1) at the row the forward passes (radices rp) leave each frequency
int rader_maps(int n, int p, const std::vector<int>& rq_, const std::vector<int>& rp_, std::vector<unsigned>& pin, std::vector<unsigned>& pout, std::vector<double>& bre, std::vector<double>& bim) {
    const int q = n / p, P1 = p - 1;
    auto powmod = [](long long b, long long e, long long m) { long long r = 1; b %= m; while (e > 0) { if (e & 1) r = r * b % m; b = b * b % m; e >>= 1; } return r; };
    int g = 0;  // the smallest generator of the units mod p
    for (int c = 2; c < p && !g; ++c) {
        bool ok = true;
        for (int f = 2; f <= P1 && ok; ++f) if (P1 % f == 0) { bool pf = true; for (int t = 2; t * t <= f; ++t) if (f % t == 0) pf = false; if (pf && powmod(c, P1 / f, p) == 1) ok = false; }
        if (ok) g = c;
    }
    if (!g) return XRFTHIP_BAD_ARG;
    std::vector<int> dlog((size_t)p, 0), gpow((size_t)P1);
    { long long v = 1; for (int m = 0; m < P1; ++m) { gpow[(size_t)m] = (int)v; dlog[(size_t)v] = m; v = v * g % p; } }
    // digit reversals of the passes along q and along p - 1
    std::vector<unsigned> revq, revp;
    DevBuf scratch;
    int rc = fastg_rev(rq_, q, scratch, revq);
    if (!rc) rc = fastg_rev(rp_, P1, scratch, revp);
    if (rc) return rc;
    // inverses for the index maps: i = n1 p + n2 q (mod n) -> n1 = i p^-1 (mod q), n2 = i q^-1 (mod p)
    long long pinv_q = 0, qinv_p = 0;
    for (int t = 0; t < q; ++t) if ((long long)t * p % q == 1 % q) { pinv_q = t; break; }
    for (int t = 0; t < p; ++t) if ((long long)t * q % p == 1) { qinv_p = t; break; }
    pin.assign((size_t)n, 0u); pout.assign((size_t)n, 0u);
    for (int i = 0; i < n; ++i) {
        const int n1 = q > 1 ? (int)((long long)i * pinv_q % q) : 0, n2 = (int)((long long)i * qinv_p % p);
        const int j = n2 == 0 ? P1 : (P1 - dlog[(size_t)n2]) % P1;  // g^-j = n2
        pin[(size_t)i] = (unsigned)(j * q + n1);
    }
    for (int k = 0; k < n; ++k) {
        const int k1 = k % q, k2 = k % p;
        const int blk = k2 == 0 ? P1 : dlog[(size_t)k2];  // the inverse passes leave X[.][g^k] at block k
        pout[(size_t)k] = (unsigned)(blk * q + (int)revq[(size_t)k1]);
    }
    const long double pi2 = 2.0L * 3.14159265358979323846264338327950288L;
    bre.assign((size_t)P1, 0.0); bim.assign((size_t)P1, 0.0);
    for (int f = 0; f < P1; ++f) {  // by the definition, in long double
        long double sr = 0, si = 0;
        for (int m = 0; m < P1; ++m) {
            const long double a = -pi2 * ((long double)gpow[(size_t)m] / (long double)p + (long double)((long long)f * m % P1) / (long double)P1);
            sr += cosl(a); si += sinl(a);
        }
        bre[(size_t)revp[(size_t)f]] = (double)(sr / P1);
        bim[(size_t)revp[(size_t)f]] = (double)(si / P1);
    }
    return XRFTHIP_OK;
}
template <typename T> int fastgy_rader_tables(xrfthip_plan* P) {
    const int n = (int)P->gy_n, p = P->gy_rad_p, P1 = p - 1;
    std::vector<unsigned> pin, pout;
    std::vector<double> bre, bim;
    int rc = rader_maps(n, p, P->g_ry, P->gy_rp, pin, pout, bre, bim);
    if (rc) return rc;
    std::vector<C2<T>> bh((size_t)P1);
    for (int f = 0; f < P1; ++f) { bh[(size_t)f].re = (T)bre[(size_t)f]; bh[(size_t)f].im = (T)bim[(size_t)f]; }
    P->g_hrevy = pout;
    rc = P->g_revy.upload(pout.data(), pout.size() * sizeof(unsigned));
    if (!rc) rc = P->gy_permin.upload(pin.data(), pin.size() * sizeof(unsigned));
    if (!rc) rc = P->gy_radb.upload(bh.data(), bh.size() * sizeof(C2<T>));
    if (!rc) rc = build_twiddle<T>(P->gy_twp, P1, P1);
    return rc;
}
// ... of the two-pass pipeline's column kernel (fastn.h, fastn_cols_kernel<T, 2, 16>): 16-bit row tables, W_q then W_(p-1) in one staged table
template <typename T> int fastn_rader_tables(xrfthip_plan* P) {
    const int n = (int)P->d.ny, p = P->n_rad_p, P1 = p - 1;
    std::vector<unsigned> pin, pout;
    std::vector<double> bre, bim;
    int rc = rader_maps(n, p, P->n_rq, P->n_rp, pin, pout, bre, bim);
    if (rc) return rc;
    const int q = n / p;
    std::vector<C2<T>> bh((size_t)P1), tw((size_t)q + P1);
    for (int f = 0; f < P1; ++f) { bh[(size_t)f].re = (T)bre[(size_t)f]; bh[(size_t)f].im = (T)bim[(size_t)f]; }
    const long double pi2 = 2.0L * 3.14159265358979323846264338327950288L;
    for (int k = 0; k < q; ++k) { const long double a = -pi2 * (long double)k / (long double)q; tw[(size_t)k].re = (T)cosl(a); tw[(size_t)k].im = (T)sinl(a); }
    for (int k = 0; k < P1; ++k) { const long double a = -pi2 * (long double)k / (long double)P1; tw[(size_t)q + k].re = (T)cosl(a); tw[(size_t)q + k].im = (T)sinl(a); }
    std::vector<uint16_t> pi16((size_t)n), po16((size_t)n);
    for (int i = 0; i < n; ++i) { pi16[(size_t)i] = (uint16_t)pin[(size_t)i]; po16[(size_t)i] = (uint16_t)pout[(size_t)i]; }
    RGeo rg{};
    rg.p = p; rg.q = n / p; rg.nrq = (int)P->n_rq.size(); rg.nrp = (int)P->n_rp.size();
    for (int i = 0; i < rg.nrq; ++i) rg.rq[i] = P->n_rq[(size_t)i];
    for (int i = 0; i < rg.nrp; ++i) rg.rp[i] = P->n_rp[(size_t)i];
    rc = P->n_c.twm.upload(tw.data(), tw.size() * sizeof(C2<T>));
    if (!rc) rc = P->n_radpin.upload(pi16.data(), pi16.size() * sizeof(uint16_t));
    if (!rc) rc = P->n_radpout.upload(po16.data(), po16.size() * sizeof(uint16_t));
    if (!rc) rc = P->n_radb.upload(bh.data(), bh.size() * sizeof(C2<T>));
    if (!rc) rc = P->n_rgeo.upload(&rg, sizeof(RGeo));
    return rc;
}
// the tables of the Bluestein form: c[k] = exp(i pi k^2 / n), k < n, and FFT_m(chirp kernel) / m at the row the forward passes leave each frequency
template <typename T> int fastgy_blue_tables(xrfthip_plan* P) {
    const long long N = P->d.ny;
    const int m = P->gy_blue_m;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<C2<T>> c((size_t)N);
    std::vector<double> br((size_t)m, 0.0), bi((size_t)m, 0.0);
    for (long long k = 0; k < N; ++k) {
        const long double a = pi * (long double)((k * k) % (2 * N)) / (long double)N;  // k^2 mod 2N keeps the angle small
        const long double cr = cosl(a), ci = sinl(a);
        c[(size_t)k].re = (T)cr; c[(size_t)k].im = (T)ci;
        br[(size_t)k] = (double)cr; bi[(size_t)k] = (double)ci;
        if (k) { br[(size_t)(m - k)] = (double)cr; bi[(size_t)(m - k)] = (double)ci; }
    }
    host_fft_smooth(br, bi);
    std::vector<C2<T>> bh((size_t)m);
    for (int k = 0; k < m; ++k) {
        bh[(size_t)P->g_hrevy[(size_t)k]].re = (T)(br[(size_t)k] / m);
        bh[(size_t)P->g_hrevy[(size_t)k]].im = (T)(bi[(size_t)k] / m);
    }
    int rc = P->gy_bluec.upload(c.data(), c.size() * sizeof(C2<T>));
    if (!rc) rc = P->gy_blueb.upload(bh.data(), bh.size() * sizeof(C2<T>));
    return rc;
}
int run_fastgy(const xrfthip_plan* P, const void* in, const void* in_b, void* out, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    FastGY p{};
    p.in = in; p.out = out;
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    p.in_b = in_b; p.two = two ? 1 : 0; p.angle = d.out_mode == XRFTHIP_OUT_PHASE ? 1 : 0;
    const bool rows = P->gy_rows;  // (the contiguous axis of a 1-D plan: "columns" are the batch's rows)
    const int ax = rows ? 1 : 0;
    p.ny = (int)P->gy_n; p.nx = rows ? (int)d.batch : (int)d.nx; p.G = P->gy_G; p.lg = ilog2i(P->gy_G);
    p.cin = P->cplx_in ? 1 : 0;
    const int ucols = ((P->cplx_in || two) ? 1 : 2) * P->gy_G;  // columns of a unit
    p.nblk = (int)((p.nx + ucols - 1) / ucols);
    p.nunits = rows ? (long long)p.nblk : d.batch * p.nblk;
    p.nry = (int)P->g_ry.size();
    for (int i = 0; i < p.nry; ++i) p.ry[i] = P->g_ry[(size_t)i];
    p.tw_y = P->g_twy.p; p.rev_y = (const unsigned*)P->g_revy.p;
    p.blue_m = P->gy_blue_m; p.blue_c = P->gy_bluec.p; p.blue_b = P->gy_blueb.p; p.tw_lds = P->gy_tw_lds ? 1 : 0;
    p.rad_p = P->gy_rad_p; p.rad_q = P->gy_rad_p ? (int)(P->gy_n / P->gy_rad_p) : 0;
    p.nrp = (int)P->gy_rp.size();
    for (int i = 0; i < p.nrp; ++i) p.rp[i] = P->gy_rp[(size_t)i];
    p.tw_p = P->gy_twp.p; p.rad_b = P->gy_radb.p; p.perm_in = (const unsigned*)P->gy_permin.p;
    p.win_y = P->win[ax].p;
    p.ph_y = P->fph[ax].p; p.ph_on = (d.out_mode != XRFTHIP_OUT_POWER && P->fph_on && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
    p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
    p.ishift_in = ((d.flags & XRFTHIP_INVERSE) && (d.flags & (rows ? XRFTHIP_ISHIFT_X : XRFTHIP_ISHIFT_Y))) ? (int)(P->gy_n / 2) : 0;
    p.ph_in = ((d.flags & XRFTHIP_PHASE_IN) && P->fph_on) ? 1 : 0;
    p.detrend = d.detrend;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0; p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.shift_y = (d.flags & (rows ? XRFTHIP_SHIFT_X : XRFTHIP_SHIFT_Y)) ? (int)(P->gy_n / 2) : 0;
    p.scale = d.scale;
    const dim3 grid((unsigned)std::min<long long>(p.nunits, 0x7fffffffLL)), blk((unsigned)P->gy_thr);
    xrfthip_plan::ProfRec* rec = prof_begin(P, rows ? "fastg_rows_rader" : "fastg_yonly", st);
#define GY_(TT, MM) do { if (P->gy_blue_m) { auto k = &fastgy_kernel<TT, MM, 1>; XRFT_LAUNCH(k, grid, blk, P->gy_lds, st, p); } \
                         else if (P->gy_rows) { auto k = &fastgy_kernel<TT, MM, 3>; XRFT_LAUNCH(k, grid, blk, P->gy_lds, st, p); } \
                         else if (P->gy_rad_p) { auto k = &fastgy_kernel<TT, MM, 2>; XRFT_LAUNCH(k, grid, blk, P->gy_lds, st, p); } \
                         else { auto k = &fastgy_kernel<TT, MM, 0>; XRFT_LAUNCH(k, grid, blk, P->gy_lds, st, p); } } while (0)
    const bool cplx = d.out_mode != XRFTHIP_OUT_POWER;  // (complex spectrum, cross spectrum, cross phase: MODE 0)
    if (P->dbl) { if (cplx) GY_(double, 0); else GY_(double, 1); } else { if (cplx) GY_(float, 0); else GY_(float, 1); }
#undef GY_
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

// threads per slab.  The passes are chains of LDS round trips, so it is the number of waves in flight on a CU that sets the rate, and that is
// bounded twice: by the registers (float32: 105 -> 4 waves per SIMD, 16 per CU; float64: 153 -> 3 and 12) and by how many slabs' LDS a CU holds.
// Take the workgroup of 1, 2, 4, 8 or 16 waves (whole waves per SIMD, or the second workgroup does not fit beside the first) that keeps most
// waves resident, the smaller one on a tie: a 50 x 50 slab is a 128-thread workgroup, eight to a CU; 96 x 96: 256 threads, four to a CU;
// 150 x 150 fills the LDS alone and brings 1024 threads (512 in float64).  Measured: profiles/r04_small_slabs.txt
long long fastg_threads(const xrfthip_plan* P) {
    const long long maxthr = P->dbl ? fastg_max_threads<double>() : fastg_max_threads<float>();
    const long long forced = env_ll("XRFTHIP_FASTG_THREADS", 0);
    if (forced >= 64 && forced <= maxthr && forced % 64 == 0) return forced;
    const long long cu_waves = P->dbl ? 12 : 16, by_lds = std::max<long long>(1, std::min<long long>(32, (long long)(kLdsMax / std::max<size_t>(1, P->g_lds))));
    long long best = 1, best_res = 0;
    for (long long w = 1; w * 64 <= maxthr; w *= 2) {
        const long long res = std::min(by_lds, cu_waves / w) * w;
        if (res > best_res) { best = w; best_res = res; }
    }
    return best * 64;
}

int run_fastg(const xrfthip_plan* P, const void* in, const void* in_b, void* out, double* iso, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    FastG p{};
    p.in = in; p.in_b = in_b; p.out = out; p.nslabs = d.batch;
    p.ny = P->g_one_d ? P->g_rows : (int)d.ny; p.nx = (int)d.nx; p.n = P->g_n; p.rs = P->g_rs; p.packed = P->g_packed ? 1 : 0;
    p.one_d = P->g_one_d ? 1 : 0; p.nrows = d.batch; p.nred = P->g_nred;
    p.cin = P->cplx_in ? 1 : 0;
    p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
    p.ishy = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_Y)) ? (int)(d.ny / 2) : 0;  // (an inverse plan rotates its fftshifted input; a forward plan folds the shift into the phase)
    p.ishx = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_X)) ? (int)(d.nx / 2) : 0;
    p.ph_in = ((d.flags & XRFTHIP_PHASE_IN) && P->fph_on) ? 1 : 0;
    p.c2r = (d.flags & XRFTHIP_C2R_X) ? 1 : 0;
    if (P->g_one_d) p.nslabs = (d.batch + P->g_rows - 1) / P->g_rows;
    p.nrx = (int)P->g_rx.size(); p.nry = (int)P->g_ry.size();
    for (int i = 0; i < p.nrx; ++i) p.rx[i] = P->g_rx[(size_t)i];
    for (int i = 0; i < p.nry; ++i) p.ry[i] = P->g_ry[(size_t)i];
    p.tw_x = P->g_twx.p; p.tw_y = P->g_twy.p; p.tw_r = P->g_twr.p;
    p.rev_x = (const unsigned*)P->g_revx.p; p.rev_y = (const unsigned*)P->g_revy.p;
    if (d.flags & XRFTHIP_ISO) {
        p.iso = iso; p.nbins = P->nbins;
        p.iso_pos = (const unsigned short*)P->g_isopos.p; p.iso_start = (const unsigned*)P->g_isostart.p;
        if (d.flags & XRFTHIP_NO_SPECTRUM_OUT) out = nullptr;
        p.out = out;
    }
    const bool win = P->win[0].p || P->win[1].p;
    p.win_y = win ? (P->win[0].p ? P->win[0].p : P->ones4096.p) : nullptr;
    p.win_x = win ? (P->win[1].p ? P->win[1].p : P->ones4096.p) : nullptr;
    const bool cplx = d.out_mode == XRFTHIP_OUT_COMPLEX, cross = d.out_mode == XRFTHIP_OUT_CROSS;
    p.ph_y = P->fph[0].p; p.ph_x = P->fph[1].p; p.ph_on = ((cplx || cross) && P->fph_on && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
    p.detrend = d.detrend;
    p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
    p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0;
    p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.scale = d.scale;
    const long long thr = fastg_threads(P);
    {   // (one_d) lanes that share a row in the per-row sums: a power of two, <= 64, <= threads / rows
        int lpr = 1;
        while (lpr * 2 <= 64 && (long long)lpr * 2 * p.ny <= thr) lpr *= 2;
        p.lpr = lpr;
    }
    const dim3 grid((unsigned)std::min<long long>(p.nslabs, 0x7fffffffLL)), blk((unsigned)thr);
    xrfthip_plan::ProfRec* rec = prof_begin(P, P->g_one_d ? "fastg_rows" : "fastg_slab", st);
#define GL_(TT, MM) do { if (P->cplx_in) { auto k = &fastg_kernel<TT, (MM == 2 ? 1 : MM), true>; XRFT_LAUNCH(k, grid, blk, P->g_lds, st, p); } \
                         else { auto k = &fastg_kernel<TT, MM, false>; XRFT_LAUNCH(k, grid, blk, P->g_lds, st, p); } } while (0)
    const bool real_out = d.out_mode == XRFTHIP_OUT_POWER || (d.flags & XRFTHIP_C2R_X);  // (MODE 1: |F|^2, or the real samples of an irfftn)
    if (P->dbl) { if (cross) GL_(double, 2); else if (!real_out) GL_(double, 0); else GL_(double, 1); }
    else { if (cross) GL_(float, 2); else if (!real_out) GL_(float, 0); else GL_(float, 1); }
#undef GL_
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}


template int fastg_setup_t<float>(xrfthip_plan*);
template int fastg_setup_t<double>(xrfthip_plan*);
template int fastgy_rader_tables<float>(xrfthip_plan*);
template int fastgy_rader_tables<double>(xrfthip_plan*);
template int fastn_rader_tables<float>(xrfthip_plan*);
template int fastn_rader_tables<double>(xrfthip_plan*);
template int fastgy_blue_tables<float>(xrfthip_plan*);
template int fastgy_blue_tables<double>(xrfthip_plan*);

// kernels of this unit that take more than 64 KB of dynamic LDS (the one-pass lengths-as-data kernels): called once through set_kernel_attrs_once()
void set_attrs_fastg() {
    const int m = (int)kLdsMax;
#define SETF(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    SETF((fastg_kernel<float, 0, false>)); SETF((fastg_kernel<float, 1, false>)); SETF((fastg_kernel<double, 0, false>)); SETF((fastg_kernel<double, 1, false>));
    SETF((fastg_kernel<float, 2, false>)); SETF((fastg_kernel<double, 2, false>));
    SETF((fastg_kernel<float, 0, true>)); SETF((fastg_kernel<float, 1, true>)); SETF((fastg_kernel<double, 0, true>)); SETF((fastg_kernel<double, 1, true>));
    SETF((fastgy_kernel<float, 0, 0>)); SETF((fastgy_kernel<float, 1, 0>)); SETF((fastgy_kernel<double, 0, 0>)); SETF((fastgy_kernel<double, 1, 0>));
    SETF((fastgy_kernel<float, 0, 1>)); SETF((fastgy_kernel<float, 1, 1>)); SETF((fastgy_kernel<double, 0, 1>)); SETF((fastgy_kernel<double, 1, 1>));
    SETF((fastgy_kernel<float, 0, 2>)); SETF((fastgy_kernel<float, 1, 2>)); SETF((fastgy_kernel<double, 0, 2>)); SETF((fastgy_kernel<double, 1, 2>));
    SETF((fastgy_kernel<float, 0, 3>)); SETF((fastgy_kernel<float, 1, 3>)); SETF((fastgy_kernel<double, 0, 3>)); SETF((fastgy_kernel<double, 1, 3>));
#undef SETF
}

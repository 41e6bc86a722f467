#include "plan.h"

// ---------------------------------------------------------------------------------------------------------------
// mixed-radix float64 form of the y-first pipeline (fastm.h)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int N, int GOV = 0> MGeomRt mgeom_t() {
    typedef MGeom<T, N, GOV> M;
    typedef typename M::template Rows<M::GR1> R1;
    return {M::THR, M::G, M::LDS, M::LDS_ROWS, M::R0, M::R1, M::R2, R1::THR, M::GR1, R1::LDS};
}
bool fastm_len(long long n, bool dbl) {
#define X_(NN) if (n == NN) return true;
    XRFT_M_LATLON(X_)
    if (dbl) { XRFT_M_POW2(X_) } else { XRFT_M_F32ONLY(X_) }
#undef X_
    return false;
}
MGeomRt mgeom(long long n, bool dbl) {
    if (dbl) {
#define X_(NN) if (n == NN) return mgeom_t<double, NN>();
        XRFT_M_LATLON(X_) XRFT_M_POW2(X_)
#undef X_
    }
#define X_(NN) if (n == NN) return mgeom_t<float, NN>();
    XRFT_M_LATLON(X_) XRFT_M_F32ONLY(X_) XRFT_M_F32_1AX(X_)
#undef X_
    return mgeom_t<float, 360>();
}
// pass 1 with four float32 sequences per workgroup (fastm_cols_kernel, GOV = 4) when the rows divide into its 8-column blocks
bool fastm_wide(long long ny, long long nx, bool dbl) {
    if (dbl || (nx & 7) != 0) return false;
#define X_(NN) if (ny == NN) return true;
    XRFT_M_WIDE32(X_)
#undef X_
    return false;
}
MGeomRt mgeom_cols(long long ny, long long nx, bool dbl) {  // geometry of pass 1 of an (ny, nx) slab
    if (fastm_wide(ny, nx, dbl)) {
#define X_(NN) if (ny == NN) return mgeom_t<float, NN, 4>();
        XRFT_M_WIDE32(X_)
#undef X_
    }
    return mgeom(ny, dbl);
}
// layout of the intermediate: CW = 2 G columns of a pass-1 workgroup, RK rows per 128-byte line
int fastm_cw(long long ny, long long nx, bool dbl) { return 2 * mgeom_cols(ny, nx, dbl).g; }
int fastm_rk(long long ny, long long nx, bool dbl) { const int lb = fastm_cw(ny, nx, dbl) * (dbl ? 16 : 8); return lb >= 128 ? 1 : 128 / lb; }
int fastm_rpu(long long nx, bool two, bool dbl) { const MGeomRt r = mgeom(nx, dbl); return two ? r.g / 2 : r.g_r1; }  // 
// rows per line of the intermediate for a (ny, nx) plan: a whole 128-byte line of pass 1's CW columns, but never more rows than
// one pass-2 workgroup owns (long float32 sequences: two per workgroup = 4 columns = 32 bytes per row, pass 2 takes 2 rows -> 64-byte pieces)
int fastm_rk2(long long ny, long long nx, bool two, bool dbl) { return std::max(1, std::min(fastm_rk(ny, nx, dbl), fastm_rpu(nx, two, dbl))); }
// ... of a plan: the table's geometry, or what fastn_setup chose when either pass runs on the run-time-radix kernels (fastn.h)
bool plan_two(const xrfthip_plan* P) { return P->d.out_mode == XRFTHIP_OUT_CROSS || P->d.out_mode == XRFTHIP_OUT_PHASE; }
int plan_cw(const xrfthip_plan* P) { return P->fastn ? P->n_cw : fastm_cw(P->yny, P->ynx, P->dbl); }
int plan_rk2(const xrfthip_plan* P) { return P->fastn ? P->n_rk : fastm_rk2(P->yny, P->ynx, plan_two(P), P->dbl); }
int plan_nxb(const xrfthip_plan* P) { return P->fastn ? P->n_nxb : (int)(P->ynx / fastm_cw(P->yny, P->ynx, P->dbl)); }

// radial sums inside pass 2 when the per-bin tables fit behind the transforms' LDS (64 KB of dynamic LDS per workgroup); otherwise
// the spectrum is stored and summed by run_radial_sums
// a radial bin map (fastm_build_tfirst) is gathered per bin without atomics or tables; a cross spectrum with a true-phase factor keeps
// the general path (the factor of a sample and of its Hermitian twin differ)
bool fastm_iso_gather(const xrfthip_plan* P) {
    return P->fastm && (P->d.flags & XRFTHIP_ISO) && P->nbins >= 1 && P->ytfirst_on && !(P->d.out_mode == XRFTHIP_OUT_CROSS && P->fph_on);
}
bool fastm_iso_fused(const xrfthip_plan* P) {
    if (!P->fastm || !(P->d.flags & XRFTHIP_ISO) || P->nbins < 1) return false;
    if (fastm_iso_gather(P)) return true;
    if (P->fastn && P->n_r.rt) return false;  // (the run-time-radix row kernel fuses the gather of a radial map only: any other map is summed from the stored spectrum)
    const bool cx = P->d.out_mode == XRFTHIP_OUT_CROSS;
    const MGeomRt R = mgeom(P->ynx, P->dbl);
    return (cx ? R.lds_rows : R.lds_r1) + (size_t)P->nbins * (cx ? 20 : 12) <= 64 * 1024;
}

// copies of the per-bin tables in pass 2 (a power of two <= 8, whatever fits the 64 KB)
int fastm_iso_ncopy(const xrfthip_plan* P) {
    if (fastm_iso_gather(P)) return 1;
    const bool cx = P->d.out_mode == XRFTHIP_OUT_CROSS;
    const MGeomRt R = mgeom(P->ynx, P->dbl);
    const size_t per = (size_t)P->nbins * (cx ? 20 : 12), room = 64 * 1024 - (cx ? R.lds_rows : R.lds_r1);
    int nc = 1;
    while (nc < 8 && per * (size_t)(2 * nc) <= room) nc *= 2;
    return nc;
}

// rows per pass-2 workgroup of this plan: two fields share a workgroup's sequences (MRowsG in fastm.h)
int fastm_gather_rpu(const xrfthip_plan* P) { return fastm_rows_rpu(P); }
int fastm_rows_rpu(const xrfthip_plan* P) {
    if (P->fastn) return P->n_rpu;
    const bool two = P->d.out_mode == XRFTHIP_OUT_CROSS || P->d.out_mode == XRFTHIP_OUT_PHASE;
    const MGeomRt r = mgeom(P->ynx, P->dbl);
    return two ? r.g / 2 : r.g_r1;
}


// ---------------------------------------------------------------------------------------------------------------
// the same pipeline with the lengths as data (fastn.h)
// ---------------------------------------------------------------------------------------------------------------
// n as a product of 2 .. kNMaxPass butterflies of fastn.h's set: the fewest passes, then the smallest largest radix (registers; threads per pass), then the
// smallest sum; ascending, so that the last pass -- one butterfly per thread -- has the fewest butterflies.  False: n has another prime factor, or too many passes.
bool fastn_factor(long long n, int maxr, std::vector<int>& out, int need_last) {  // need_last: the largest radix must reach it (the last pass: one butterfly per thread)
    static const int R[] = {20, 18, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    static const int PR[] = {2, 3, 5, 7, 11, 13};
    int ex[6] = {0, 0, 0, 0, 0, 0};
    long long m = n;
    for (int i = 0; i < 6; ++i) while (m % PR[i] == 0) { ++ex[i]; m /= PR[i]; }
    if (m != 1 || n < 4) return false;
    int re[17][6];
    for (int i = 0; i < 17; ++i) { int v = R[i]; for (int k = 0; k < 6; ++k) { re[i][k] = 0; while (v % PR[k] == 0) { ++re[i][k]; v /= PR[k]; } } }
    std::vector<int> best, cur;
    auto better = [](const std::vector<int>& a, const std::vector<int>& b) {  // (a complete, b the incumbent)
        if (b.empty()) return true;
        if (a.size() != b.size()) return a.size() < b.size();
        const int ma = *std::max_element(a.begin(), a.end()), mb = *std::max_element(b.begin(), b.end());
        if (ma != mb) return ma < mb;
        int sa = 0, sb = 0; for (int v : a) sa += v; for (int v : b) sb += v;
        return sa < sb;
    };
    std::function<void(int)> dfs = [&](int from) {
        bool done = true;
        for (int k = 0; k < 6; ++k) if (ex[k]) done = false;
        if (done) { if (cur.size() >= 2 && cur[0] >= need_last && better(cur, best)) best = cur; return; }
        if ((int)cur.size() >= kNMaxPass || (!best.empty() && cur.size() + 1 > best.size()) || (!cur.empty() && cur[0] < need_last)) return;  // (non-increasing: cur[0] is the largest)
        for (int i = from; i < 17; ++i) {  // non-increasing radices: each multiset once
            if (R[i] > maxr) continue;
            bool fits = true;
            for (int k = 0; k < 6; ++k) if (re[i][k] > ex[k]) fits = false;
            if (!fits) continue;
            for (int k = 0; k < 6; ++k) ex[k] -= re[i][k];
            cur.push_back(R[i]);
            dfs(i);
            cur.pop_back();
            for (int k = 0; k < 6; ++k) ex[k] += re[i][k];
        }
    };
    dfs(0);
    if (best.empty()) return false;
    std::sort(best.begin(), best.end());
    out = best;
    return true;
}

// the geometry of one n-point transform held in LDS with g sequences per workgroup (fastn.h, NGeo); blue: the Bluestein plan's natural layout is its intermediate layout
void fastn_geom(long long n, const std::vector<int>& rad, int g, int maxthr, bool blue, NGeo& o, int thr_force, int thr_pref) {
    o = NGeo{};
    o.n = (int)n; o.np = (int)rad.size();
    long long L = n;
    for (int p = 0; p < o.np; ++p) { o.r[p] = rad[(size_t)p]; o.inv_r[p] = 1.0f / (float)rad[(size_t)p]; o.m[p] = (int)(L / rad[(size_t)p]); L /= rad[(size_t)p]; }
    const int rl = o.r[o.np - 1], pdq = (rl % 2 == 0) ? rl : 0;
    int pnq = (o.r[0] % 2 == 0) ? o.r[0] : 0;
    if (blue) pnq = pdq;
    o.inv_pdq = pdq ? 1.0f / (float)pdq : 0.0f;
    o.inv_pnq = pnq ? 1.0f / (float)pnq : 0.0f;
    o.pn_r0 = (pnq != 0 && pnq == o.r[0]) ? 1 : 0;
    for (int p = 0; p < o.np; ++p) o.step[p] = o.m[p] + ((pdq && p + 1 < o.np) ? o.m[p] / pdq : 0);
    o.wlast = 1;
    for (int p = 1; p + 1 < o.np; ++p) o.wlast *= o.r[p];
    int acc = 0;
    for (int p = 1; p + 1 < o.np; ++p) { o.two[p] = acc; acc += o.m[p] * o.r[p]; }
    if (blue) { o.two[0] = acc; acc += o.m[0]; }  // (W_n^j, j < m[0]: the first pass of a Bluestein plan runs from LDS, too)
    o.twn = acc;
    const long long span = n + std::max<long long>(pdq ? n / pdq : 0, pnq ? n / pnq : 0) + 1;
    o.str = (int)(((span + 3) / 8) * 8 + 4);  // the smallest s >= span with s = 4 (mod 8): sequences eight lanes touch land on disjoint banks (fastm.h)
    o.g = g; o.lg = ilog2i(g);
    long long bmax = 0;
    for (int p = 0; p < o.np; ++p) bmax = std::max<long long>(bmax, n / o.r[p]);
    long long thr = 0;
    (void)bmax; (void)thr_pref;
    const long long lower = ((g * (n / rl) + 63) / 64) * 64;  // (the last pass: one butterfly per thread)
    thr = std::max<long long>(lower, std::min<long long>(maxthr, ((std::max(thr_force, 64) + 63) / 64) * 64));
    o.thr = (int)thr;
}

template <typename T> int fastn_upload_twm(const NGeo& g, DevBuf& buf, bool blue) {  // W_{L_p}^(j k) at [two[p] + j r[p] + k], p = 1 .. np - 2
    std::vector<C2<T>> t((size_t)std::max(g.twn, 1));
    const long double pi2 = 2.0L * 3.14159265358979323846264338327950288L;
    if (blue)
        for (int j = 0; j < g.m[0]; ++j) {
            const long double a = -pi2 * (long double)j / (long double)g.n;
            t[(size_t)(g.two[0] + j)].re = (T)cosl(a); t[(size_t)(g.two[0] + j)].im = (T)sinl(a);
        }
    for (int p = 1; p + 1 < g.np; ++p) {
        const int Lp = g.m[p] * g.r[p];
        for (int j = 0; j < g.m[p]; ++j)
            for (int k = 0; k < g.r[p]; ++k) {
                const long double a = -pi2 * (long double)(((long long)j * k) % Lp) / (long double)Lp;
                t[(size_t)(g.two[p] + j * g.r[p] + k)].re = (T)cosl(a);
                t[(size_t)(g.two[p] + j * g.r[p] + k)].im = (T)sinl(a);
            }
    }
    return buf.upload(t.data(), t.size() * sizeof(C2<T>));
}

// Bluestein tables of pass 1: c[k] = exp(i pi k^2 / n), k < n, and FFT_m(chirp) / m in natural order
template <typename T> int fastn_blue_tables(xrfthip_plan* P) {
    const long long N = P->d.ny;
    const int m = P->n_blue_m;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<C2<T>> c((size_t)N), bh((size_t)m);
    std::vector<double> br((size_t)m, 0.0), bi((size_t)m, 0.0);
    for (long long k = 0; k < N; ++k) {
        const long double a = pi * (long double)((k * k) % (2 * N)) / (long double)N;
        const long double cr = cosl(a), ci = sinl(a);
        c[(size_t)k].re = (T)cr; c[(size_t)k].im = (T)ci;
        br[(size_t)k] = (double)cr; bi[(size_t)k] = (double)ci;
        if (k) { br[(size_t)(m - k)] = (double)cr; bi[(size_t)(m - k)] = (double)ci; }
    }
    host_fft_smooth(br, bi);
    for (int k = 0; k < m; ++k) { bh[(size_t)k].re = (T)(br[(size_t)k] / m); bh[(size_t)k].im = (T)(bi[(size_t)k] / m); }
    int rc = P->n_bluec.upload(c.data(), c.size() * sizeof(C2<T>));
    if (!rc) rc = P->n_blueb.upload(bh.data(), bh.size() * sizeof(C2<T>));
    return rc;
}

size_t fastn_lds(const NGeo& g, size_t csize, bool cols) {
    return ((size_t)g.g * g.str + g.twn) * csize + (cols ? (size_t)(g.thr / 64) * g.g * 4 * sizeof(double) : 0);
}

// Radices and thread count of one transform with g sequences per workgroup.  What counts is how many workgroups a CU keeps resident, and that is set by the
// registers (128 per lane in float32 -> 16 waves per CU, 168 in float64 -> 12): the thread count is a divisor of that budget -- 512 (columns) / 256 (rows) in
// float32, 192 / 256 / 384 in float64; 576- or 320-thread workgroups leave a CU half empty (profiles/r05_fastn_threads.txt) -- and the factorisation is the one
// with the fewest passes whose LAST radix is large enough for one last-pass butterfly per thread at that count (a thread loops over the other passes' butterflies).
size_t fastn_lds(const NGeo& g, size_t csize, bool cols);
bool fastn_pick(long long n, int g, bool blue, bool dbl, bool cols, int maxr, int thr_force, NGeo& out) {
    const int maxthr = dbl ? fastn_max_threads<double>() : fastn_max_threads<float>();
    std::vector<int> base, r;
    if (!fastn_factor(n, maxr, base)) return false;
    static const int kD[] = {192, 256, 384, 512, 0}, kFC[] = {512, 256, 1024, 0, 0}, kFR[] = {256, 512, 1024, 0, 0};
    const int* targets = dbl ? kD : (cols && !blue) ? kFC : kFR;  // (a chirp convolution: the smaller workgroup, more of them)
    const int budget = dbl ? 12 : 16;  // waves a CU keeps resident at the kernels' register counts
    for (int extra = 0; extra <= 1; ++extra) {
        int best_res = -1;
        for (int i = 0; i < 5 && (thr_force > 0 ? i < 1 : targets[i] != 0); ++i) {
            const int t = thr_force > 0 ? std::min(maxthr, ((thr_force + 63) / 64) * 64) : targets[i];
            const int need = (int)((g * n + t - 1) / t);
            if (need > maxr || !fastn_factor(n, maxr, r, need) || r.size() > base.size() + (size_t)extra) continue;
            NGeo cand{};
            fastn_geom(n, r, g, maxthr, blue, cand, t);
            const size_t lds = fastn_lds(cand, dbl ? 16 : 8, cols);
            if (lds > 156 * 1024) continue;
            const int w = cand.thr / 64, res = std::min<int>(budget / w, (int)((160 * 1024) / lds)) * w;  // resident waves per CU
            // float64: the size that keeps the most waves resident ((64, 1440, 720): 256 threads, three workgroups by LDS, 5.4 us against 6.6 with 192); float32:
            // the first size that works -- 512 (columns) / 256 (rows): beyond that a workgroup that owns the CU's LDS alone only gets slower (2200-point
            // columns: 512 threads 137 GFFT/s, 1024 threads 122)
            if (res > best_res) { best_res = res; out = cand; }
            if (!dbl) break;
        }
        if (best_res >= 0) return true;
    }
    return false;
}

bool rader_split(long long n, bool allow17, int& p_out, std::vector<int>& rq, std::vector<int>& rp);
// Decide which kernel runs each pass of a y-first plan on (ny, nx) and the layout of the intermediate between them.  Returns false when the plan stays
// with the other paths (a length the butterflies do not factor and the chirp convolution does not fit, sequences that do not fit the LDS).
bool fastn_setup(xrfthip_plan* P) {
    const xrfthip_desc& d = P->d;
    const bool dbl = P->dbl, two = plan_two(P);
    const size_t cs = P->csize;
    const int maxthr = dbl ? fastn_max_threads<double>() : fastn_max_threads<float>();
    const int maxr = (int)env_ll("XRFTHIP_FASTN_MAXR", dbl ? fastn_max_radix<double>() : fastn_max_radix<float>());
    if (d.ny < 16 || d.nx < 16 || d.ny > 16384 || d.nx > 16384 || (unsigned long long)d.ny * (unsigned long long)d.nx * P->rsize >= (1ULL << 32)) return false;
    const bool tab_ok = env_ll("XRFTHIP_FASTN_TABLES", 1) != 0;  // (0: the run-time-radix kernels even where the table has the length -- measurements)
    bool cols_rt = !(tab_ok && fastm_len(d.ny, dbl)), rows_rt = !(tab_ok && fastm_len(d.nx, dbl));
    if (!cols_rt && d.nx % fastm_cw(d.ny, d.nx, dbl) != 0) cols_rt = true;  // (the table's column kernel wants whole column blocks: (180, 180) float64 -- 8-column blocks -- took the generic passes)
    if (!cols_rt && !rows_rt) return false;  // (plain fastm)
    // ---- rows (length nx)
    std::vector<int> rx, ry;
    int rpu = 0;
    NGeo gr{};
    if (rows_rt) {
        if (!fastn_factor(d.nx, maxr, rx)) return false;
        // rows per workgroup: reads and writes are contiguous whatever the count, and many small workgroups interleave their phases best (fastm.h): the
        // count that leaves 6, else 3, 2, 1 workgroups on a CU -- but two rows at least while they fit, so that W2's lines hold two rows' pieces
        const long long forced = env_ll("XRFTHIP_FASTN_RPU", 0);
        // (measured, profiles/r05_fastn_threads.txt: two rows per workgroup -- whole 128-byte lines of W2 -- beat one and four at every size, even where
        // two rows leave a single workgroup on a CU: (16, 3000, 3000) float64 78 against 66 GFFT/s)
        // SHORT rows (nx <= 512: the 73 x 144, 37 x 72, 145 x 192 grids that the Rader columns brought here): two 144-point rows are a 288-point workgroup, 150 000 of
        // them per call -- as many rows as make ~1152 points (8 at most), and one thread per ~9 points: (4096, 73, 144) float32 rows 287 -> 86 us, (16384, 37, 72)
        // 507 -> 106 (profiles/r05_small_awkward.txt)
        int rpu_short = 2, thr_short = 0;
        if (d.nx <= 512 && !two) {
            while (rpu_short < 8 && (long long)rpu_short * 2 * d.nx <= 1152) rpu_short *= 2;
            const long long pts = (long long)rpu_short * d.nx;
            thr_short = pts < 1024 ? 64 : pts < 2304 ? 128 : 0;
        }
        static const size_t caps[] = {156 * 1024};
        for (int ci = 0; ci < 1 && !rpu; ++ci)
            for (int cand = forced ? 16 : rpu_short; cand >= 1 && !rpu; cand >>= 1) {
                if (forced && cand != forced) continue;
                NGeo t{};
                const int tr_env = (int)env_ll("XRFTHIP_FASTN_TR", 0), tr = tr_env ? tr_env : (cand == rpu_short && !forced) ? thr_short : 0;
                if (!(tr && fastn_pick(d.nx, two ? 2 * cand : cand, false, dbl, false, maxr, tr, t)) && !fastn_pick(d.nx, two ? 2 * cand : cand, false, dbl, false, maxr, tr_env, t)) continue;
                if ((long long)t.g * (d.nx / t.r[t.np - 1]) > maxthr) continue;
                if (fastn_lds(t, cs, false) <= caps[ci] && 2 * cand <= 64) { rpu = cand; gr = t; }
            }
        if (!rpu) return false;
    } else {
        rpu = fastm_rpu(d.nx, two, dbl);
        if (rpu < 1) return false;
    }
    // ---- columns (length ny, or the chirp convolution's m)
    int cw = 0, blue_m = 0, rad_p = 0;
    std::vector<int> rq, rp;
    NGeo gc{};
    if (cols_rt) {
        long long mlen = d.ny;
        if (!fastn_factor(d.ny, maxr, ry) && d.ny <= 8192 && env_ll("XRFTHIP_FASTN_RADER", 1) && rader_split(d.ny, true, rad_p, rq, rp)) {
            // ONE prime factor 17 ... 127 with a smooth p - 1 (721 = 7 x 103 latitudes, 365 = 5 x 73): the prime-factor form with Rader's algorithm along the prime
            // inside the column tile (fastg.h, fastn_cols_kernel<T, 2, 16>): the tile is [ny][G], no padding; ~2.4 transforms of the length in LDS where the chirp
            // convolution takes two of 2.1 x the length.  Column pairs per workgroup and threads as for the chirp convolution: small workgroups, several per CU
            const int gmax_r = dbl ? 4 : 8;
            int G = 0;
            NGeo t{};
            const long long forced = env_ll("XRFTHIP_FASTN_GC", 0), thr_f = env_ll("XRFTHIP_FASTN_TC", 0);
            // (measured, profiles/r05_rader_cols.txt: the widest block of ~3000 ... 6000 points -- (365, 720) float32 8 pairs 65 us against 87 with 4, 721 points 4 pairs,
            // 1460 points 4 pairs and 512 threads 379 us against 430 with 256; float64 (365, 720) 4 pairs 112 us against 167 with 2)
            static const int kOrd[] = {8, 4, 2, 1};
            static const size_t caps[] = {52 * 1024, 78 * 1024, 156 * 1024};
            for (int ci = 0; ci < 3 && !G; ++ci)
                for (int oi = 0; oi < 4 && !G; ++oi) {
                    const int cand = kOrd[oi];
                    if (cand > gmax_r || (forced && cand != forced)) continue;
                    if (!forced && cand > 1 && (long long)cand * d.ny > 6000) continue;
                    if (!rows_rt && d.nx % (2 * cand) != 0) continue;
                    if (2LL * cand > d.nx + 1) continue;
                    NGeo c{};
                    c.n = (int)d.ny; c.np = 0; c.g = cand; c.lg = ilog2i(cand); c.str = (int)d.ny; c.twn = (int)(d.ny / rad_p) + rad_p - 1;
                    // (threads by the points of a workgroup: 73 x 8 pairs on 64 threads 163 us against 327 on 256 -- a single wave has no barriers to wait at)
                    const long long pts = (long long)cand * d.ny;
                    c.thr = thr_f ? (int)std::min<long long>(maxthr, (thr_f + 63) / 64 * 64) : pts <= 1536 ? 64 : pts <= 2560 ? 128 : (pts >= 4096 && !dbl) ? 512 : 256;
                    const size_t lds = fastn_lds(c, cs, true) + 2 * (((size_t)d.ny + 7) & ~(size_t)7) * 2;
                    if (lds <= caps[ci]) { G = cand; t = c; }
                }
            if (G) {
                gc = t; cw = 2 * G;
            } else rad_p = 0;
        }
        if (rad_p) {
        } else if (!fastn_factor(d.ny, maxr, ry)) {
            // a prime factor without a butterfly: x conj(c) zero-padded to m >= 2 ny - 1 -> FFT_m -> * FFT_m(chirp) / m -> inverse FFT_m -> * conj(c).  The m with the
            // fewest passes within 12 % of the smallest candidate
            std::vector<int> best;
            long long bm = 0;
            double bcost = 0.0;
            // (arithmetic of an r-point butterfly per point, roughly: the prime butterflies 7 / 11 / 13 are O(r^2))
            static const double kFlop[21] = {0, 0, 2, 5, 4, 8, 8, 15, 8, 10, 12, 24, 11, 28, 19, 15, 11, 0, 14, 0, 15};
            for (long long m = 2 * d.ny - 1; m <= (2 * d.ny - 1) * 9 / 8 + 16; ++m) {
                std::vector<int> t;
                if (!fastn_factor(m, std::min(maxr, 16), t)) continue;
                double c = 0.0;
                for (int r : t) c += 12.0 + kFlop[r];  // (a trip through LDS + the butterfly, per point and pass)
                c *= (double)m;
                if (best.empty() || t.size() < best.size() || (t.size() == best.size() && c < bcost)) { best = t; bm = m; bcost = c; }
            }
            if (best.empty()) return false;
            ry = best; mlen = bm; blue_m = (int)bm;
        }
        const int gmax = dbl ? 4 : 8, gpref = dbl ? 2 : 4;  // (32-byte row segments at least where they fit: 16-byte segments load at half the rate, fastm.h)
        const long long forced = env_ll("XRFTHIP_FASTN_GC", 0);
        // sequences per workgroup: the widest row segments (32 bytes at least where they fit: 16-byte segments load at half the rate, fastm.h) that leave three,
        // else two, else one workgroup on a CU; a chirp convolution -- bound by its 16 trips through the LDS, not by its loads -- the narrowest
        // instead: more, smaller workgroups interleave better ((64, 721, 1440): 2 pairs x 256 threads 99 GFFT/s, 4 x 512 80; profiles/r05_fastn_knobs.txt)
        int G = 0;
        static const size_t caps[] = {52 * 1024, 78 * 1024, 156 * 1024};
        static const int kBlueOrder[] = {2, 4, 1, 8}, kOrder[] = {8, 4, 2, 1};
        for (int ci = 0; ci < 3 && !G; ++ci)
            for (int oi = 0; oi < 4 && !G; ++oi) {
                // (a SHORT chirp convolution -- 94 x 192, 181 x 360, 241 x 480 grids: m < 1024 -- takes the widest block of <= 2048 points like everything else here:
                // (2048, 94, 192) 8 pairs on 128 threads 197 us against 617 with 2 on 256, (1024, 181, 360) 496 against 935; profiles/r05_chirp_small.txt)
                const bool blue_short = blue_m && mlen < 1024 && !dbl;
                const int cand = (blue_m && !blue_short) ? kBlueOrder[oi] : kOrder[oi];
                if (blue_short && !forced && cand > 1 && (long long)cand * mlen > 2048) continue;
                if (cand > gmax) continue;
                if (forced && cand != forced) continue;
                if (ci < 2 && cand < gpref && !forced && !blue_m) continue;
                if (!rows_rt && d.nx % (2 * cand) != 0) continue;  // (the table's row kernel reads an unpadded intermediate)
                if (2LL * cand > d.nx + 1) continue;
                NGeo t{};
                // threads by the points of the column block: SHORT columns ((512, 100, 2000): 8 pairs = 800 points) on the 512 threads of the large slabs leave most
                // waves idle at every barrier -- 64 threads 275 us against 728, (1024, 98, 1000) 336 against 1295; 2000 ... 4000 points: 256 (profiles/r05_short_cols.txt)
                const long long pts = (long long)cand * mlen;
                const int tc_env = (int)env_ll("XRFTHIP_FASTN_TC", 0);
                // (float64: 64 threads up to 768 points, 128 up to 2048 -- (128, 500, 1500) 473 us against 575 with 192, (128, 250, 3000) 406 against 521)
                const int tc = tc_env ? tc_env : blue_short ? (pts <= 1024 ? 64 : pts <= 2560 ? 128 : 256) : blue_m ? 0
                               : pts <= (dbl ? 768 : 1536) ? 64 : (dbl && pts <= 2048) ? 128 : (!dbl && pts < 4096) ? 256 : 0;
                const int mr = blue_m ? std::min(maxr, 16) : maxr;
                // (the last pass wants one butterfly per thread: where the count is too small for the radices at hand, the next one up)
                bool picked = false;
                for (int tt = tc; tt && tt <= 256 && !picked && !tc_env; tt *= 2) picked = fastn_pick(mlen, cand, blue_m != 0, dbl, true, mr, tt, t);
                if (!picked && !fastn_pick(mlen, cand, blue_m != 0, dbl, true, mr, tc_env, t)) continue;
                if ((long long)t.g * (mlen / t.r[t.np - 1]) > maxthr) continue;
                if (fastn_lds(t, cs, true) <= caps[ci]) { G = cand; gc = t; }
            }
        if (!G && !rad_p) return false;
        if (!rad_p) cw = 2 * G;
    } else {
        cw = fastm_cw(d.ny, d.nx, dbl);
    }
    const int nxb = (int)((d.nx + cw - 1) / cw);
    const long long pitch = (long long)nxb * cw;
    if (!rows_rt && pitch != d.nx) return false;
    int rk = (int)std::max<long long>(1, std::min<long long>((long long)(128 / (cw * cs)), rpu));
    if (rpu % rk != 0) return false;
    P->fastn = true;
    P->n_c.rt = cols_rt; P->n_c.geo = gc; P->n_c.lds = cols_rt ? fastn_lds(gc, cs, true) + (rad_p ? 2 * (((size_t)d.ny + 7) & ~(size_t)7) * 2 : 0) : 0;
    P->n_rad_p = rad_p; P->n_rq = rq; P->n_rp = rp;
    P->n_dbg = (int)env_ll("XRFTHIP_FASTN_DBG", 0);
    P->n_r.rt = rows_rt; P->n_r.geo = gr; P->n_r.lds = rows_rt ? fastn_lds(gr, cs, false) : 0;
    P->n_cw = cw; P->n_rk = rk; P->n_rpu = rpu; P->n_nxb = nxb; P->y_pitch = pitch; P->n_blue_m = blue_m;
    return true;
}

FastN fastn_wrap(const xrfthip_plan* P, const FastM& m, bool cols) {
    FastN n{};
    n.f = m;
    n.g = (NGeoPtr)(cols ? P->n_c.geo_dev.p : P->n_r.geo_dev.p);
    n.twm = cols ? P->n_c.twm.p : P->n_r.twm.p;
    n.pitch = (int)P->y_pitch; n.nxb = P->n_nxb;
    n.pair_ok = (P->ynx % 2 == 0) ? 1 : 0;
    n.blue_c = P->n_bluec.p; n.blue_b = P->n_blueb.p;
    n.rg = (RGeoPtr)P->n_rgeo.p; n.rad_pin = (const unsigned short*)P->n_radpin.p; n.rad_pout = (const unsigned short*)P->n_radpout.p; n.rad_b = P->n_radb.p;
    const bool cplx_out = P->d.out_mode == XRFTHIP_OUT_COMPLEX || P->d.out_mode == XRFTHIP_OUT_CROSS;
    const int vw = (int)(16 / (cplx_out ? P->csize : P->rsize));
    n.vec_ok = (P->ynx % vw == 0) ? 1 : 0;
    n.rpu = P->n_rpu;
    n.dbg = P->n_dbg;
    return n;
}

void fastn_launch_cols(const xrfthip_plan* P, const FastM& m, hipStream_t st) {
    const FastN n = fastn_wrap(P, m, true);
    const NGeo& hg = P->n_c.geo;
    const dim3 grid((unsigned)(8 * ((m.nunits + 7) / 8))), blk((unsigned)hg.thr);
    const size_t lds = P->n_c.lds;
    int maxrad = 0;
    for (int i = 0; i < hg.np; ++i) maxrad = std::max(maxrad, hg.r[i]);
#define NC_(TT, CC) do { if (P->n_blue_m) { auto k = &fastn_cols_kernel<TT, 1, 16>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } /* (a chirp convolution's radices stop at 16) */ \
                         else if (P->n_rad_p) { auto k = &fastn_cols_kernel<TT, 2, 16>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } \
                         else { auto k = &fastn_cols_kernel<TT, 0, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } } while (0)
    if (P->dbl) NC_(double, 16); else if (maxrad > 16) NC_(float, 20); else NC_(float, 16);
#undef NC_
}

void fastn_launch_rows(const xrfthip_plan* P, const FastM& m, long long gc, bool fused, hipStream_t st) {
    const FastN n = fastn_wrap(P, m, false);
    const xrfthip_desc& d = P->d;
    const NGeo& hg = P->n_r.geo;
    const dim3 grid((unsigned)(gc * (P->y_nrow_pad / P->n_rpu))), blk((unsigned)hg.thr);
    const size_t lds = P->n_r.lds;
    int maxrad = 0;
    for (int i = 0; i < hg.np; ++i) maxrad = std::max(maxrad, hg.r[i]);
#define NR_(TT, CC) do { \
        if (d.out_mode == XRFTHIP_OUT_POWER) { if (fused) { auto k = &fastn_rows_kernel<TT, 1, true, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } else { auto k = &fastn_rows_kernel<TT, 1, false, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } } \
        else if (d.out_mode == XRFTHIP_OUT_CROSS) { if (fused) { auto k = &fastn_rows_kernel<TT, 2, true, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } else { auto k = &fastn_rows_kernel<TT, 2, false, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } } \
        else if (d.out_mode == XRFTHIP_OUT_PHASE) { auto k = &fastn_rows_kernel<TT, 3, false, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } \
        else { auto k = &fastn_rows_kernel<TT, 0, false, CC>; XRFT_LAUNCH(k, grid, blk, lds, st, n); } } while (0)
    if (P->dbl) NR_(double, 16); else if (maxrad > 16) NR_(float, 20); else NR_(float, 16);
#undef NR_
}

FastM fastm_params(const xrfthip_plan* P, const void* in, void* out, char* ws, long long g0, long long gc, int slot, long long slot_slabs) {
    const xrfthip_desc& d = P->d;
    const size_t slab_pts = (size_t)P->yny * P->ynx, s0 = (size_t)slot * slot_slabs;
    FastM p{};
    p.in = (const char*)in + (size_t)g0 * slab_pts * P->rsize;
    p.w2 = ws + P->off_w + s0 * (size_t)P->y_nrow_pad * (size_t)P->y_pitch * P->csize;
    const size_t out_esz = (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_PHASE) ? P->rsize : P->csize;
    const size_t out_pts = (size_t)P->yny * ((d.flags & XRFTHIP_HALF_X) ? P->ynx / 2 + 1 : P->ynx);
    p.out = out ? (char*)out + (size_t)g0 * out_pts * out_esz : nullptr;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0;
    p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.tw_x = P->tw_fx.p; p.tw_y = P->tw_fy.p;
    p.win_y = P->win[0].p ? P->win[0].p : P->ones4096.p;
    p.win_x = P->win[1].p ? P->win[1].p : P->ones4096.p;
    p.colfit = reinterpret_cast<double*>(ws + P->off_rowfit) + s0 * (size_t)P->ynx * 4;
    p.corr = ws + P->off_corr + s0 * (size_t)P->ynx * P->csize;
    p.ph_y = P->fph[0].p; p.ph_x = P->fph[1].p; p.ph_on = P->fph_on ? 1 : 0;
    p.what0 = P->ywhat0.p; p.what1 = P->ywhat1.p;
    p.binmap = (const int*)P->binmap.p; p.nbins = P->nbins; p.iso_ncopy = P->nbins > 0 ? fastm_iso_ncopy(P) : 1;
    p.iso_part = reinterpret_cast<double*>(ws + P->off_isopart);
    const bool gather = fastm_iso_gather(P);
    p.tfirst = gather ? reinterpret_cast<const unsigned short*>(P->ytfirst.p) : nullptr;
    p.twin = gather ? reinterpret_cast<const unsigned*>(P->ytwin.p) : nullptr;
    p.ny = (int)P->yny; p.nx = (int)P->ynx; p.nrow_pad = P->y_nrow_pad;
    p.l_cw = ilog2i(plan_cw(P)); p.l_rk = ilog2i(plan_rk2(P));
    p.detrend = d.detrend; p.nslab = (int)gc;
    p.nunits = (int)(gc * plan_nxb(P));
    p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(P->yny / 2) : 0;
    p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(P->ynx / 2) : 0;
    p.scale = d.scale;
    return p;
}

void fastm_launch_cols(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const MGeomRt C = mgeom_cols(P->yny, P->ynx, P->dbl);
    const bool wide = fastm_wide(P->yny, P->ynx, P->dbl);
    const bool rt = P->fastn && P->n_c.rt;  // (the run-time-radix kernel: fastn.h)
    xrfthip_plan::ProfRec* rec = prof_begin(P, rt ? "fastn_cols" : "fastm_cols", st);
    if (rt) fastn_launch_cols(P, p, st);
    const dim3 grid((unsigned)(8 * ((p.nunits + 7) / 8))), blk((unsigned)C.thr);
#ifdef XRFT_M_BIGLDS  /* profiling builds with more than 64 KB of LDS per workgroup */
#define MBIG_(k, n) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(n))
#else
#define MBIG_(k, n) ((void)0)
#endif
#define MC_(TT, NN) do { if (d.detrend) { auto k = &fastm_cols_kernel<TT, NN, true>; MBIG_(k, C.lds_cols); XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } \
                         else { auto k = &fastm_cols_kernel<TT, NN, false>; MBIG_(k, C.lds_cols); XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } } while (0)
#define XD_(NN) if (P->yny == NN) MC_(double, NN);
#define XF_(NN) if (P->yny == NN) MC_(float, NN);
#define MCW_(NN) if (P->yny == NN) do { if (d.detrend) { auto k = &fastm_cols_kernel<float, NN, true, 4>; MBIG_(k, C.lds_cols); XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } \
                                            else { auto k = &fastm_cols_kernel<float, NN, false, 4>; MBIG_(k, C.lds_cols); XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } } while (0);
    if (rt) {}
    else if (wide) { XRFT_M_WIDE32(MCW_) }
    else if (P->dbl) { XRFT_M_LATLON(XD_) XRFT_M_POW2(XD_) } else { XRFT_M_LATLON(XF_) XRFT_M_F32ONLY(XF_) }
#undef MCW_
#undef XD_
#undef XF_
#undef MC_
    prof_end(rec, st);
    if (d.detrend) {
        rec = prof_begin(P, "fastm_fit", st);
        if (P->dbl) {
            auto kf = &fastm_fit_kernel<double>;
            XRFT_LAUNCH(kf, dim3((unsigned)gc), dim3(256), 3 * 256 * sizeof(double), st, (const double*)p.colfit, (const double*)p.win_x,
                        reinterpret_cast<C2<double>*>(const_cast<void*>(p.corr)), (int)P->ynx, (int)P->yny, (int)d.detrend);
        } else {
            auto kf = &fastm_fit_kernel<float>;
            XRFT_LAUNCH(kf, dim3((unsigned)gc), dim3(256), 3 * 256 * sizeof(double), st, (const double*)p.colfit, (const float*)p.win_x,
                        reinterpret_cast<C2<float>*>(const_cast<void*>(p.corr)), (int)P->ynx, (int)P->yny, (int)d.detrend);
        }
        prof_end(rec, st);
    }
}

void fastm_launch_rows(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const MGeomRt R = mgeom(P->ynx, P->dbl);
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    const bool rt = P->fastn && P->n_r.rt;  // (the run-time-radix kernel: fastn.h)
    xrfthip_plan::ProfRec* rec = prof_begin(P, rt ? "fastn_rows" : "fastm_rows", st);
    const bool fused = fastm_iso_fused(P), full = two;  // (full: pass 1's sequence count per workgroup)
    if (rt) { fastn_launch_rows(P, p, gc, fused, st); prof_end(rec, st); return; }
    const dim3 grid((unsigned)(gc * (P->y_nrow_pad / fastm_rows_rpu(P)))), blk((unsigned)(full ? R.thr : R.thr_r1));
    const size_t lds_rows = full ? R.lds_rows : R.lds_r1;
    const size_t lds_iso = p.tfirst ? lds_rows : lds_rows + (size_t)P->nbins * (d.out_mode == XRFTHIP_OUT_CROSS ? 20 : 12) * (size_t)p.iso_ncopy;  // (the gather needs no tables)
#define MR_(TT, NN) do { \
        if (d.out_mode == XRFTHIP_OUT_POWER) { if (fused) { auto k = &fastm_rows_kernel<TT, NN, 1, true>; XRFT_LAUNCH(k, grid, blk, lds_iso, st, p); } else { auto k = &fastm_rows_kernel<TT, NN, 1>; MBIG_(k, lds_rows); XRFT_LAUNCH(k, grid, blk, lds_rows, st, p); } } \
        else if (d.out_mode == XRFTHIP_OUT_CROSS) { if (fused) { auto k = &fastm_rows_kernel<TT, NN, 2, true>; XRFT_LAUNCH(k, grid, blk, lds_iso, st, p); } else { auto k = &fastm_rows_kernel<TT, NN, 2>; XRFT_LAUNCH(k, grid, blk, R.lds_rows, st, p); } } \
        else if (d.out_mode == XRFTHIP_OUT_PHASE) { auto k = &fastm_rows_kernel<TT, NN, 3>; XRFT_LAUNCH(k, grid, blk, R.lds_rows, st, p); } \
        else { auto k = &fastm_rows_kernel<TT, NN, 0>; XRFT_LAUNCH(k, grid, blk, lds_rows, st, p); } } while (0)
#define XD_(NN) if (P->ynx == NN) MR_(double, NN);
#define XF_(NN) if (P->ynx == NN) MR_(float, NN);
    if (P->dbl) { XRFT_M_LATLON(XD_) XRFT_M_POW2(XD_) } else { XRFT_M_LATLON(XF_) XRFT_M_F32ONLY(XF_) }
#undef XD_
#undef XF_
#undef MR_
    prof_end(rec, st);
}

int run_fastm(const xrfthip_plan* P, const void* in, const void* in1, void* out, double* iso, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    const bool iso_on = (d.flags & XRFTHIP_ISO) != 0;
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        FastM p = fastm_params(P, in, out, ws, g0, gc, 0, P->G);
        const bool fused = fastm_iso_fused(P);
        if (iso_on && !out && !fused) p.out = ws + P->off_isotmp;  // radial sums from the stored spectrum: the group's spectrum lives in the workspace
        fastm_launch_cols(P, p, gc, st);
        if (two) {
            const FastM p1 = fastm_params(P, in1, out, ws, g0, gc, 1, P->G);
            fastm_launch_cols(P, p1, gc, st);
            p.w2b = p1.w2;
            p.corr_b = p1.corr;
        }
        fastm_launch_rows(P, p, gc, st);
        HIP_TRY(hipGetLastError());
        if (iso_on && fused) {  // the row workgroups' partial sums, added in order
            const int hw = d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1, nb = P->nbins * hw, upr = P->y_nrow_pad / fastm_rows_rpu(P);
            xrfthip_plan::ProfRec* rec = prof_begin(P, "iso_reduce", st);
            auto kr = &iso_reduce_kernel;
            XRFT_LAUNCH(kr, dim3((unsigned)((nb + 63) / 64), (unsigned)gc), dim3(256), 4 * 64 * sizeof(double), st, (const double*)p.iso_part,
                        iso + (size_t)g0 * nb, upr, nb, p.tfirst ? reinterpret_cast<const unsigned*>(P->ytunits.p) : nullptr, hw);
            prof_end(rec, st);
            HIP_TRY(hipGetLastError());
        } else if (iso_on) {  // radial sums of the stored spectrum (xrft.py:895-906), bit-reproducible
            const bool cx = d.out_mode == XRFTHIP_OUT_CROSS;
            xrfthip_plan::ProfRec* rec = prof_begin(P, "radial_sums", st);
            const int rc = run_radial_sums(cx ? (P->dbl ? XRFTHIP_C128 : XRFTHIP_C64) : (P->dbl ? XRFTHIP_F64 : XRFTHIP_F32), p.out, (const int32_t*)P->binmap.p, gc, d.ny, d.nx, p.shift_y, p.shift_x, P->nbins,
                                           P->iso_chunks, reinterpret_cast<double*>(ws + P->off_isopart), iso + (size_t)g0 * P->nbins * (cx ? 2 : 1), st);
            prof_end(rec, st);
            if (rc) return rc;
        }
    }
    return XRFTHIP_OK;
}

// one transform axis, not the contiguous one: pass 1 alone (fastm_yonly_kernel)
bool fastmy_len(long long n, bool dbl) {
#define X_(NN) if (n == NN) return true;
    XRFT_M_LATLON(X_) XRFT_M_POW2(X_) XRFT_M_YONLY(X_)
    if (!dbl) { XRFT_M_F32ONLY(X_) XRFT_M_F32_1AX(X_) }
#undef X_
    return n == 2048 || n == 4096;
}
template <typename T, int N> MGeomRt mygeom_t() {  // (the y-only kernel's own geometry: at least two sequences per workgroup)
    typedef typename MYGeom<T, N>::type M;
    return {M::THR, M::G, M::LDS, M::LDS_ROWS, M::R0, M::R1, M::R2, 0, 0, 0};
}
MGeomRt mygeom(long long n, bool dbl) {
    if (n == 4096) return dbl ? mygeom_t<double, 4096>() : mygeom_t<float, 4096>();
    if (n == 2048 && dbl) return mygeom_t<double, 2048>();
    if (!dbl) {
#define X_(NN) if (n == NN) return mgeom_t<float, NN>();
        XRFT_M_POW2(X_) XRFT_M_YONLY(X_) X_(2048)
#undef X_
    } else {
#define X_(NN) if (n == NN) return mgeom_t<double, NN>();
        XRFT_M_YONLY(X_)
#undef X_
    }
    return mgeom(n, dbl);
}
int run_fastmy(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const MGeomRt C = mygeom(d.ny, P->dbl);
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    FastM p{};
    p.in = in; p.in_b = in1; p.out = out;
    p.angle = d.out_mode == XRFTHIP_OUT_PHASE ? 1 : 0;
    p.tw_y = P->tw_fy.p;
    p.win_y = P->win[0].p ? P->win[0].p : P->ones4096.p;
    p.ph_y = P->fph[0].p; p.ph_on = (P->fph_on && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
    p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
    p.ishift_in = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_Y)) ? (int)(d.ny / 2) : 0;
    p.ph_in = ((d.flags & XRFTHIP_PHASE_IN) && P->fph_on) ? 1 : 0;
    p.ny = (int)d.ny; p.nx = (int)d.nx;
    p.detrend = d.detrend; p.nslab = (int)d.batch;
    p.cin = P->cplx_in ? 1 : 0;
    p.nunits = (int)(d.batch * ((P->cplx_in && !two) ? (d.nx + C.g - 1) / C.g : d.nx / (two ? C.g : 2 * C.g)));  // (complex columns: the last block of a slab may be short)
    p.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0; p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.scale = d.scale;
    xrfthip_plan::ProfRec* rec = prof_begin(P, "fastm_yonly", st);
    const dim3 grid((unsigned)(8 * ((p.nunits + 7) / 8))), blk((unsigned)C.thr);
#define MYL_(TT, NN, MM) do { auto k = &fastm_yonly_kernel<TT, NN, MM>; XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } while (0)
#define MY_(TT, NN) do { \
        if (d.out_mode == XRFTHIP_OUT_POWER) MYL_(TT, NN, 1); else if (two) MYL_(TT, NN, 2); else MYL_(TT, NN, 0); } while (0)
#define XD_(NN) if (d.ny == NN) MY_(double, NN);
#define XF_(NN) if (d.ny == NN) MY_(float, NN);
    if (P->dbl) { XRFT_M_LATLON(XD_) XRFT_M_POW2(XD_) XRFT_M_YONLY(XD_) XD_(2048) XD_(4096) } else { XRFT_M_LATLON(XF_) XRFT_M_F32ONLY(XF_) XRFT_M_F32_1AX(XF_) XRFT_M_POW2(XF_) XRFT_M_YONLY(XF_) XF_(2048) XF_(4096) }
#undef XD_
#undef XF_
#undef MY_
#undef MYL_
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

// one transform axis, the contiguous one, short rows: rows packed in pairs (fastm_xonly_kernel).  Rows are contiguous whatever the
// number of sequences per workgroup, so the lengths that leave room for one pair only (4096; 2048 in float64) are taken too.
bool fastmx_len(long long n, bool dbl) { return fastmy_len(n, dbl); }
MGeomRt mxgeom(long long n, bool dbl) {
    if (n == 4096) return dbl ? mgeom_t<double, 4096>() : mgeom_t<float, 4096>();
    if (dbl && n == 2048) return mgeom_t<double, 2048>();
    return mygeom(n, dbl);
}
int run_fastmx(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const MGeomRt C = mxgeom(d.nx, P->dbl);
    const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    FastM p{};
    p.in = in; p.in_b = in1; p.out = out;
    p.angle = d.out_mode == XRFTHIP_OUT_PHASE ? 1 : 0;
    p.tw_x = P->tw_fx.p;
    p.win_x = P->win[1].p ? P->win[1].p : P->ones4096.p;
    p.ph_x = P->fph[1].p; p.ph_on = (P->fph_on && !(d.flags & XRFTHIP_PHASE_IN)) ? 1 : 0;
    p.inv = (d.flags & XRFTHIP_INVERSE) ? 1 : 0;
    p.ishift_in = ((d.flags & XRFTHIP_INVERSE) && (d.flags & XRFTHIP_ISHIFT_X)) ? (int)(d.nx / 2) : 0;
    p.ph_in = ((d.flags & XRFTHIP_PHASE_IN) && P->fph_on) ? 1 : 0;
    p.ny = 1; p.nx = (int)d.nx;
    p.detrend = d.detrend; p.nslab = (int)d.batch;
    p.half = (d.flags & XRFTHIP_HALF_X) ? 1 : 0;
    p.realdim2 = (d.flags & XRFTHIP_REALDIM_X2) ? 1 : 0;
    p.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
    p.scale = d.scale;
    xrfthip_plan::ProfRec* rec = prof_begin(P, "fastm_xonly", st);
    p.cin = P->cplx_in ? 1 : 0;
    p.c2r = (P->cplx_in && (d.flags & XRFTHIP_C2R_X)) ? 1 : 0;  // (irfft rows: two half rows per transform)
    const int rpw = (two || (P->cplx_in && !p.c2r)) ? C.g : 2 * C.g;
    const dim3 grid((unsigned)((d.batch + rpw - 1) / rpw)), blk((unsigned)C.thr);
#define MXL_(TT, NN, MM) do { auto k = &fastm_xonly_kernel<TT, NN, MM>; XRFT_LAUNCH(k, grid, blk, C.lds_cols, st, p); } while (0)
#define MX_(TT, NN) do { \
        if (d.out_mode == XRFTHIP_OUT_POWER) MXL_(TT, NN, 1); else if (two) MXL_(TT, NN, 2); else MXL_(TT, NN, 0); } while (0)
#define XD_(NN) if (d.nx == NN) MX_(double, NN);
#define XF_(NN) if (d.nx == NN) MX_(float, NN);
    if (P->dbl) { XRFT_M_LATLON(XD_) XRFT_M_POW2(XD_) XRFT_M_YONLY(XD_) XD_(2048) XD_(4096) } else { XRFT_M_LATLON(XF_) XRFT_M_F32ONLY(XF_) XRFT_M_F32_1AX(XF_) XRFT_M_POW2(XF_) XRFT_M_YONLY(XF_) XF_(2048) XF_(4096) }
#undef XD_
#undef XF_
#undef MX_
#undef MXL_
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}


template int fastn_upload_twm<float>(const NGeo&, DevBuf&, bool);
template int fastn_upload_twm<double>(const NGeo&, DevBuf&, bool);
template int fastn_blue_tables<float>(xrfthip_plan*);
template int fastn_blue_tables<double>(xrfthip_plan*);

// kernels of this unit that take more than 64 KB of dynamic LDS (the mixed-radix and run-time-radix kernels): called once through set_kernel_attrs_once()
void set_attrs_fastm() {
    const int m = (int)kLdsMax;
#define SETF(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    // (the one fastm instantiation above 64 KB of dynamic LDS: 4096-point float64 rows, one pair per workgroup)
#define YA_(TT, NN) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, m); \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, m); \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    YA_(float, 4096); YA_(double, 2048); YA_(double, 4096);
#undef YA_
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
#define SETN(TT, CC) SETF((fastn_cols_kernel<TT, 0, CC>)); SETF((fastn_cols_kernel<TT, 1, 16>)); SETF((fastn_cols_kernel<TT, 2, 16>)); \
                     SETF((fastn_rows_kernel<TT, 0, false, CC>)); SETF((fastn_rows_kernel<TT, 1, false, CC>)); SETF((fastn_rows_kernel<TT, 1, true, CC>)); SETF((fastn_rows_kernel<TT, 2, false, CC>)); \
                     SETF((fastn_rows_kernel<TT, 2, true, CC>)); SETF((fastn_rows_kernel<TT, 3, false, CC>))
    SETN(float, 16); SETN(float, 20); SETN(double, 16);
    SETF((fastn_irows_kernel<float, 0, 16>)); SETF((fastn_irows_kernel<float, 1, 16>)); SETF((fastn_irows_kernel<float, 0, 20>)); SETF((fastn_irows_kernel<float, 1, 20>));
    SETF((fastn_irows_kernel<double, 0, 16>)); SETF((fastn_irows_kernel<double, 1, 16>)); SETF((fastn_irows_kernel<double, 2, 16>));
    SETF((fastn_irows_kernel<float, 2, 16>)); SETF((fastn_irows_kernel<float, 2, 20>));
#undef SETN
#undef SETF
}

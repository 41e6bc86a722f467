// xrft_hip.cpp -- the plan builder, the generic tile passes, and the C ABI of the plan (see plan.h for the map of the host units).
#include "plan.h"

thread_local int xrfth::g_last_hip_error = 0;

namespace {

struct TileChoice { int T, threads, seq_stride, pad_shift; size_t lds; };

// pick sequences-per-tile for an n-point FFT; `col`: the tile axis is the contiguous one in memory, so T*csize
// bytes per row segment should reach a 128-byte line.  Returns T = 0 if one sequence does not fit in LDS.
TileChoice choose_tile(long long n_logical, size_t csize, bool col, long long avail, size_t hist_bytes) {
    const long long n = lds_fft_len(n_logical, csize);
    TileChoice c{};
    c.pad_shift = csize == 8 ? 4 : 3;
    long long ss = n + (n >> c.pad_shift) + 1;
    if ((ss & 1) == 0) ++ss;
    c.seq_stride = (int)ss;
    const size_t per = (size_t)ss * csize;
    const size_t hard = kLdsMax - hist_bytes - 64;
    // measured (scripts/prof_generic3.py): for long sequences larger tiles (wider chunks, more threads) beat two small
    // workgroups per CU; short ones already get 8+ sequences into 64 KiB
    const size_t soft = (size_t)env_ll("XRFTHIP_LDS_SOFT", (64 * 1024) / per < 8 ? 144 * 1024 : 64 * 1024);
    long long target = std::max<long long>(1, 8192 / std::max<long long>(n, 1));
    if (col) target = std::max<long long>(target, (long long)(128 / csize));
    long long T = std::min<long long>(target, (long long)(soft / per));
    const long long want = col ? std::min<long long>(4, target) : 1;
    if (T < want) T = std::min<long long>(target, (long long)(hard / per));
    if (T > avail) T = avail;
    if (T < 1) { c.T = 0; return c; }
    if (col) { long long p2 = 1; while (p2 * 2 <= T) p2 *= 2; T = p2; }
    c.T = (int)T;
    long long th = (T * n + 15) / 16;
    th = ((th + 63) / 64) * 64;
    c.threads = (int)std::min<long long>(csize == 16 ? 512 : 1024, std::max<long long>(64, th));  // float64: 256 VGPRs per lane
    c.lds = (((size_t)T * per + 15) & ~(size_t)15) + hist_bytes;
    return c;
}

// split n = n1 * n2 with both factors as close to sqrt(n) as the factorisation allows
bool split_two(long long n, long long& n1, long long& n2) {
    long long best = 0;
    for (long long a = 1; a * a <= n; ++a)
        if (n % a == 0) best = a;
    if (best <= 1) return false;
    n1 = n / best;  // n1 >= n2
    n2 = best;
    return true;
}

template <typename T>
struct Builder {
    xrfthip_plan& P;
    bool cur_raw = false;  // building the field-0 pipeline of a cross spectrum: its own flip flags (xrft.py:436-441 flips each field by its own coordinate)
    explicit Builder(xrfthip_plan& p) : P(p) {}
    bool flip_x() const { return (P.d.flags & (cur_raw ? XRFTHIP_FLIP0_X : XRFTHIP_FLIP_X)) != 0; }
    bool flip_y() const { return (P.d.flags & (cur_raw ? XRFTHIP_FLIP0_Y : XRFTHIP_FLIP_Y)) != 0; }

    int tables_for(int n, FftTables** out) {
        auto it = P.tables.find(n);
        if (it == P.tables.end()) {
            FftTables& t = P.tables[n];
            int rc = build_tables<T>(t, n);
            if (rc) { P.tables.erase(n); return rc; }
            *out = &t;
        } else *out = &it->second;
        return XRFTHIP_OK;
    }

    int set_fft(Pass& ps, int n) {
        FftTables* t;
        int rc = tables_for(n, &t);
        if (rc) return rc;
        ps.g.n = t->blue_m ? t->blue_m : n;
        ps.g.blue_n = t->blue_m ? n : 0;
        ps.g.blue_c = t->blue_c.p;
        ps.g.blue_b = t->blue_b.p;
        ps.g.nr = (int)t->radix.size();
        for (int i = 0; i < ps.g.nr; ++i) ps.g.radix[i] = t->radix[i];
        ps.g.tw = t->tw.p;
        ps.g.rev = (const unsigned*)t->rev.p;
        ps.generic = t->generic;
        return XRFTHIP_OK;
    }

    void apply_tile(Pass& ps, const TileChoice& c) {
        ps.g.T = c.T;
        ps.g.dbg = (int)env_ll("XRFTHIP_DBG", 0);
        ps.g.seq_stride = c.seq_stride;
        ps.g.pad_shift = c.pad_shift;
        ps.threads = c.threads;
        ps.lds = c.lds;
        // twiddle table of this pass' FFT length in LDS when it fits next to the tile
        const size_t twb = (size_t)ps.g.n * P.csize + 32;
        if (ps.g.n > 1 && twb <= 48 * 1024 && ps.lds + twb <= kLdsMax && env_ll("XRFTHIP_TW_LDS", 1)) {
            ps.g.tw_lds = 1;
            ps.lds += twb;
        }
        const size_t rvb = (size_t)ps.g.n * sizeof(unsigned) + 32;
        if (ps.g.n > 1 && !ps.g.blue_n && rvb <= 32 * 1024 && ps.lds + rvb <= kLdsMax && env_ll("XRFTHIP_REV_LDS", 1)) {
            ps.g.rev_lds = 1;
            ps.lds += rvb;
        }
    }

    void fill_prologue(Pass& ps, long long rows, long long jmp, long long jmq) {
        const xrfthip_desc& d = P.d;
        Prologue& pr = ps.pr;
        pr.in_complex = P.cplx_in;
        pr.detrend = d.detrend != XRFTHIP_DETREND_NONE;
        pr.rows = rows;
        pr.j_mul_p = jmp;
        pr.j_mul_q = jmq;
        pr.ny = (int)d.ny;
        pr.nx = (int)d.nx;
        pr.flip_y = flip_y();
        pr.ishift_y = !!(d.flags & XRFTHIP_ISHIFT_Y);
        pr.flip_x = flip_x();
        pr.ishift_x = !!(d.flags & XRFTHIP_ISHIFT_X);
        pr.slab_stride = d.ny * d.nx;
        pr.row_stride = d.nx;
        pr.conj_in = !!(d.flags & XRFTHIP_INVERSE);
        if (d.flags & XRFTHIP_C2R_X) {
            pr.herm_nxh = (int)(d.nx / 2 + 1);
            pr.row_stride = d.nx / 2 + 1;
            pr.slab_stride = d.ny * (d.nx / 2 + 1);
        }
        ps.first = true;
        ps.in_kind = B_IN;
    }

    // raw: final pass of the F0 pipeline of CROSS (plain complex spectrum, unshifted, into the F0 buffer)
    void fill_epilogue(Pass& ps, bool raw, int p_axis, long long odiv) {
        const xrfthip_desc& d = P.d;
        Epilogue& ep = ps.ep;
        ep.mode = raw ? 0 : d.out_mode;
        ep.p_axis = p_axis;
        ep.odiv = odiv;
        ep.r_mul = odiv > 1 ? 1 : 0;
        ep.q_mul = 0;
        ep.p_mul = odiv;
        ep.ny = (int)d.ny;
        ep.nx = (int)d.nx;
        ep.scale = raw ? 1.0 : d.scale;
        if (raw) {
            ep.nx_out = (int)P.width;
            ep.mirror = 0;
            ep.shift_y = ep.shift_x = 0;
            ep.realdim_x2 = 0;
            ep.slab_stride = d.ny * P.width;
            ep.row_stride = P.width;
            ps.out_kind = B_F0;
        } else {
            ep.nx_out = (int)P.nx_out;
            ep.mirror = P.mirror;
            ep.shift_y = !!(d.flags & XRFTHIP_SHIFT_Y);
            ep.shift_x = !!(d.flags & XRFTHIP_SHIFT_X);
            ep.realdim_x2 = !!(d.flags & XRFTHIP_REALDIM_X2);
            ep.conj_out = !!(d.flags & XRFTHIP_INVERSE);
            ep.real_out = !!(d.flags & XRFTHIP_C2R_X);
            ep.slab_stride = d.ny * P.nx_out;
            ep.row_stride = P.nx_out;
            ep.other_slab_stride = d.ny * P.width;
            ep.other_row_stride = P.width;
            ps.out_kind = B_OUT;
        }
        ps.final_ = true;
    }

    size_t hist_bytes(bool) const { return 0; }  // (radial sums are a pass of their own over the stored spectrum: run_radial_sums)

    // ---------------------------------------------------------------- x passes (along the contiguous axis)
    // rows_per_slab = ny (2-D) or 1 (1-D).  `last`: the x transform is the whole transform (1-D).
    int build_x(std::vector<Pass>& out, bool raw) {
        const xrfthip_desc& d = P.d;
        const bool last = d.ndim == 1;
        const long long rows = d.ny;
        const bool real_in = !P.cplx_in;
        const bool want_r2c = real_in && d.nx % 2 == 0 && d.nx >= 2 && P.width == d.nx / 2 + 1;
        const long long n = want_r2c ? d.nx / 2 : d.nx;
        TileChoice c = choose_tile(n, P.csize, false, std::max<long long>(1, rows * d.batch), last ? hist_bytes(raw) : 0);
        const bool four = P.width == d.nx && (c.T == 0 || n >= env_ll("XRFTHIP_X_FOURSTEP_MIN", 1LL << 40)) && n > 1;
        if (c.T == 0 && !four) return XRFTHIP_UNSUPPORTED_LENGTH;
        if (!four) {
            Pass ps;
            ps.label = want_r2c ? "x:r2c-row" : "x:row";
            int rc = set_fft(ps, (int)n);
            if (rc) return rc;
            apply_tile(ps, c);
            ps.g.r2c = want_r2c;
            ps.g.n_out = (int)P.width;
            if (want_r2c) {
                DevBuf* b = new DevBuf();
                P.extra.push_back(b);
                rc = build_twiddle<T>(*b, d.nx, n + 1);
                if (rc) return rc;
                ps.g.tw_r2c = b->p;
            }
            ps.g.tile_axis = 0;
            ps.g.in_fast = 0;
            ps.g.out_fast = 0;
            ps.g.inner = 1;
            ps.g.tiles_per_outer = 1;
            ps.outer_per_slab = rows;
            fill_prologue(ps, rows, 1, 0);
            if (real_in && !flip_x() && !(d.flags & (XRFTHIP_C2R_X | XRFTHIP_INVERSE)) && env_ll("XRFTHIP_LEAN_ROWS", 1) &&
                ps.lds + 48 * (size_t)c.T + 32 <= kLdsMax) {  // lean row loader: per-row constants behind everything else
                ps.g.rowc_off = (int)((ps.lds + 15) & ~(size_t)15);
                ps.lds = (size_t)ps.g.rowc_off + 48 * (size_t)c.T;
            }
            if (last) {
                fill_epilogue(ps, raw, 0, 1);
            } else {
                ps.g.out_so = P.width;
                ps.g.out_sq = 0;
                ps.g.out_sp = 1;
                ps.out_kind = B_W;
            }
            out.push_back(ps);
            return XRFTHIP_OK;
        }
        // ---- four-step along x: nx = n1 * n2, A: FFT over i1 (stride n2) + twiddle, B: FFT over i2, transposed store
        long long n1, n2;
        if (!split_two(d.nx, n1, n2)) return XRFTHIP_UNSUPPORTED_LENGTH;
        TileChoice ca = choose_tile(n1, P.csize, true, n2, 0);
        TileChoice cb = choose_tile(n2, P.csize, true, n1, last ? hist_bytes(raw) : 0);
        if (ca.T == 0 || cb.T == 0) return XRFTHIP_UNSUPPORTED_LENGTH;
        DevBuf* big = new DevBuf();
        P.extra.push_back(big);
        int rc = build_twiddle<T>(*big, d.nx, d.nx);
        if (rc) return rc;
        {
            Pass a;
            a.label = "x:four-step-A";
            rc = set_fft(a, (int)n1);
            if (rc) return rc;
            apply_tile(a, ca);
            a.g.n_out = (int)n1;
            a.g.tile_axis = 1;
            a.g.in_fast = 1;
            a.g.out_fast = 1;
            a.g.inner = n2;
            a.g.tiles_per_outer = (n2 + ca.T - 1) / ca.T;
            a.outer_per_slab = rows;
            fill_prologue(a, rows, n2, 1);
            a.g.out_so = d.nx; a.g.out_sq = 1; a.g.out_sp = n2;
            a.g.tw_big = big->p; a.g.tw_bigN = d.nx; a.g.tw_qdiv = 1; a.g.tw_qmod = n2;
            a.out_kind = B_W2;
            if (ca.T <= 64 && (ca.T & (ca.T - 1)) == 0 && a.threads % ca.T == 0 && env_ll("XRFTHIP_LEAN_COL", 1)) {
                a.g.lean_col = 2;
                if (d.ndim == 1 && real_in && !flip_x() && !(d.flags & (XRFTHIP_C2R_X | XRFTHIP_INVERSE))) a.g.lean_col |= 1;
            }
            out.push_back(a);
        }
        {
            Pass b;
            b.label = "x:four-step-B";
            rc = set_fft(b, (int)n2);
            if (rc) return rc;
            apply_tile(b, cb);
            b.g.n_out = (int)n2;
            b.g.tile_axis = 1;
            b.g.in_fast = 0;
            b.g.out_fast = 1;
            b.g.inner = n1;
            b.g.tiles_per_outer = (n1 + cb.T - 1) / cb.T;
            b.outer_per_slab = rows;
            b.in_kind = B_W2;
            b.g.in_so = d.nx; b.g.in_sq = n2; b.g.in_sp = 1;
            if (last) {
                fill_epilogue(b, raw, 0, 1);
                b.ep.q_mul = 1;  // kx = k1 + n1 * k2
                b.ep.p_mul = n1;
                if ((b.ep.mode == 0 || b.ep.mode == 1) && !b.ep.conj_out && !b.ep.real_out && !b.ep.mirror && cb.T <= 64 &&
                    (cb.T & (cb.T - 1)) == 0 && b.threads % cb.T == 0 && env_ll("XRFTHIP_LEAN_FINAL", 1))
                    b.g.lean_final = 2;
            } else {
                b.g.out_so = d.nx; b.g.out_sq = 1; b.g.out_sp = n1;
                b.out_kind = B_W;
            }
            out.push_back(b);
        }
        return XRFTHIP_OK;
    }

    // ---------------------------------------------------------------- y passes (strided axis of the intermediate)
    int build_y(std::vector<Pass>& out, bool raw) {
        const xrfthip_desc& d = P.d;
        const long long ny = d.ny, w = P.width;
        TileChoice c = choose_tile(ny, P.csize, true, w, hist_bytes(raw));
        const long long min_t = std::min<long long>(env_ll("XRFTHIP_Y_MIN_T", 4), w);
        bool four = (c.T < min_t || ny >= env_ll("XRFTHIP_Y_FOURSTEP_MIN", 1LL << 40)) && ny > 3;
        long long n1 = 0, n2 = 0;
        if (four && !split_two(ny, n1, n2)) four = false;
        if (!four) {
            if (c.T == 0) return XRFTHIP_UNSUPPORTED_LENGTH;
            Pass ps;
            ps.label = "y:col";
            int rc = set_fft(ps, (int)ny);
            if (rc) return rc;
            apply_tile(ps, c);
            ps.g.n_out = (int)ny;
            ps.g.tile_axis = 1;
            ps.g.in_fast = 1;
            ps.g.out_fast = 1;
            ps.g.inner = w;
            ps.g.tiles_per_outer = (w + c.T - 1) / c.T;
            ps.outer_per_slab = 1;
            ps.in_kind = B_W;
            ps.g.in_so = ny * w; ps.g.in_sq = 1; ps.g.in_sp = w;
            fill_epilogue(ps, raw, 1, 1);
            if ((ps.ep.mode == 0 || ps.ep.mode == 1) && !ps.ep.conj_out && !ps.ep.real_out && c.T >= 1 && c.T <= 64 &&
                (c.T & (c.T - 1)) == 0 && ps.threads % c.T == 0 && env_ll("XRFTHIP_LEAN_FINAL", 1))
                ps.g.lean_final = 1;
            out.push_back(ps);
            return XRFTHIP_OK;
        }
        TileChoice ca = choose_tile(n1, P.csize, true, n2 * w, 0);
        TileChoice cb = choose_tile(n2, P.csize, true, w, hist_bytes(raw));
        if (ca.T == 0 || cb.T == 0) return XRFTHIP_UNSUPPORTED_LENGTH;
        DevBuf* big = new DevBuf();
        P.extra.push_back(big);
        int rc = build_twiddle<T>(*big, ny, ny);
        if (rc) return rc;
        {
            Pass a;  // in place on W viewed as [slab][n1][n2*w]
            a.label = "y:four-step-A";
            rc = set_fft(a, (int)n1);
            if (rc) return rc;
            apply_tile(a, ca);
            a.g.n_out = (int)n1;
            a.g.tile_axis = 1;
            a.g.in_fast = 1;
            a.g.out_fast = 1;
            a.g.inner = n2 * w;
            a.g.tiles_per_outer = (n2 * w + ca.T - 1) / ca.T;
            a.outer_per_slab = 1;
            a.in_kind = B_W;
            a.out_kind = B_W;
            a.g.in_so = ny * w; a.g.in_sq = 1; a.g.in_sp = n2 * w;
            a.g.out_so = ny * w; a.g.out_sq = 1; a.g.out_sp = n2 * w;
            a.g.tw_big = big->p; a.g.tw_bigN = ny; a.g.tw_qdiv = w; a.g.tw_qmod = n2;
            out.push_back(a);
        }
        {
            Pass b;  // sequences (slab, k1, kx): o = slab*n1 + k1, q = kx, points i2 (stride w)
            b.label = "y:four-step-B";
            rc = set_fft(b, (int)n2);
            if (rc) return rc;
            apply_tile(b, cb);
            b.g.n_out = (int)n2;
            b.g.tile_axis = 1;
            b.g.in_fast = 1;
            b.g.out_fast = 1;
            b.g.inner = w;
            b.g.tiles_per_outer = (w + cb.T - 1) / cb.T;
            b.outer_per_slab = n1;
            b.in_kind = B_W;
            b.g.in_so = n2 * w; b.g.in_sq = 1; b.g.in_sp = w;
            fill_epilogue(b, raw, 1, n1);
            out.push_back(b);
        }
        return XRFTHIP_OK;
    }

    // XRFTHIP_AXIS_Y: one column pass that is first AND final -- reads the caller's [slab][ny][nx] array with the prologue
    // (point p = row, q = column) and writes the result in the same layout
    int build_yonly(std::vector<Pass>& out, bool raw) {
        const xrfthip_desc& d = P.d;
        TileChoice c = choose_tile(d.ny, P.csize, true, d.nx, 0);
        if (c.T == 0) return XRFTHIP_UNSUPPORTED_LENGTH;  // (longer columns: transpose on the caller's side and use a 1-D plan)
        Pass ps;
        ps.label = "y:col-only";
        int rc = set_fft(ps, (int)d.ny);
        if (rc) return rc;
        apply_tile(ps, c);
        ps.g.n_out = (int)d.ny;
        ps.g.tile_axis = 1;
        ps.g.in_fast = 1;
        ps.g.out_fast = 1;
        ps.g.inner = d.nx;
        ps.g.tiles_per_outer = (d.nx + c.T - 1) / c.T;
        ps.outer_per_slab = 1;
        fill_prologue(ps, 1, 0, 0);
        ps.pr.p_is_row = 1;
        fill_epilogue(ps, raw, 1, 1);
        out.push_back(ps);
        return XRFTHIP_OK;
    }

    int build_pipeline(std::vector<Pass>& out, bool raw) {
        cur_raw = raw;
        if (P.d.flags & XRFTHIP_AXIS_Y) return build_yonly(out, raw);
        int rc = build_x(out, raw);
        if (rc) return rc;
        if (P.d.ndim == 2) rc = build_y(out, raw);
        if (rc) return rc;
        // one row pass feeding one column pass: hand the intermediate over in tiles of the column pass's T columns
        if (out.size() == 2 && out[0].g.rowc_off > 0 && out[0].out_kind == B_W && out[1].in_kind == B_W && out[1].g.tile_axis == 1 &&
            out[1].g.T >= 2 && (out[1].g.T & (out[1].g.T - 1)) == 0 && env_ll("XRFTHIP_TILED_W", 1)) {
            const int tc = out[1].g.T;
            const long long wc = (P.width + tc - 1) / tc * tc;
            if (P.w_cols == 0 || P.w_cols == wc) {  // (the F0 pipeline of a cross spectrum picks the same T)
                P.w_cols = wc;
                out[0].g.out_tiled = tc; out[1].g.in_tiled = tc;
                out[0].g.til_stride = out[1].g.til_stride = P.d.ny * tc;
                out[0].g.til_slab = out[1].g.til_slab = (wc / tc) * P.d.ny * tc;
                out[0].g.til_ny = out[1].g.til_ny = (int)P.d.ny;
            }
        }
        // single-purpose kernel instantiations where every precondition of a lean path is known now
        if (env_ll("XRFTHIP_PATHS", 1) && !env_ll("XRFTHIP_DBG", 0)) {
            const xrfthip_desc& d = P.d;
            const bool no_iso_out = !(d.flags & (XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT)), no_phase_in = !(d.flags & XRFTHIP_PHASE_IN);
            for (Pass& ps : out) {
                if (ps.generic) continue;
                if (ps.first && ps.g.tile_axis == 0 && ps.g.rowc_off > 0 && no_phase_in && (!ps.final_ || no_iso_out)) ps.path = 1;
                else if (!ps.first && ps.final_ && ps.g.in_tiled && ps.g.lean_final == 1 && (raw || no_iso_out)) ps.path = 2;
                else if (ps.first && !ps.final_ && ps.g.lean_col == 3 && no_phase_in) ps.path = 3;
                else if (!ps.first && ps.final_ && ps.g.lean_final == 2 && (raw || no_iso_out)) ps.path = 4;
            }
        }
        return XRFTHIP_OK;
    }
};

template <typename T>
int build_plan_t(xrfthip_plan& P) {
    Builder<T> B(P);
    P.w_cols = 0;
    int rc = B.build_pipeline(P.passes, false);
    if (rc) return rc;
    if (P.d.out_mode == XRFTHIP_OUT_CROSS || P.d.out_mode == XRFTHIP_OUT_PHASE) rc = B.build_pipeline(P.passes_f0, true);
    return rc;
}

void set_kernel_attrs_once() {
    static bool done = false;
    if (done) return;
    done = true;
    const int m = (int)kLdsMax;
    // the generic tile kernels live in this unit; every family sets its own (host_*.cpp)
#define SETA(TT, A, B, C) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_fft_kernel<TT, A, B, C, sizeof(TT) == 8 ? 512 : 1024, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, m)
#define SETP(TT, A, B, PP) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_fft_kernel<TT, A, B, false, sizeof(TT) == 8 ? 512 : 1024, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, m)
#define SETPATHS(TT) SETP(TT, true, false, 1); SETP(TT, true, true, 1); SETP(TT, false, true, 2); SETP(TT, true, false, 3); SETP(TT, false, true, 4)
    SETPATHS(float);
    SETPATHS(double);
#undef SETPATHS
#undef SETP
#define SETALL(TT) SETA(TT, false, false, false); SETA(TT, false, false, true); SETA(TT, false, true, false); SETA(TT, false, true, true); \
                   SETA(TT, true, false, false); SETA(TT, true, false, true); SETA(TT, true, true, false); SETA(TT, true, true, true)
    SETALL(float);
    SETALL(double);
#undef SETALL
    set_attrs_fasty();
    set_attrs_fastm();
    set_attrs_fastg();
    set_attrs_rows();
}

template <typename T>
void launch_tile(const Pass& ps, int grid, hipStream_t st) {
    const dim3 g((unsigned)grid), b((unsigned)ps.threads);
    // float64 plans run at most 512 threads per block (choose_tile) on the instantiation with 256 VGPRs per lane; float32
    // keeps the 128-VGPR one (its small tiles want four workgroups per CU)
    constexpr int MT = sizeof(T) == 8 ? 512 : 1024;
#define LP_(A, B, PP) do { auto k = &tile_fft_kernel<T, A, B, false, MT, PP>; XRFT_LAUNCH(k, g, b, ps.lds, st, ps.g, ps.pr, ps.ep); } while (0)
    if (ps.path && !ps.generic) {  // single-purpose instantiations (preconditions checked when the plan was built)
        if (ps.path == 1 && ps.first && !ps.final_) { LP_(true, false, 1); return; }
        if (ps.path == 1 && ps.first && ps.final_) { LP_(true, true, 1); return; }
        if (ps.path == 2 && !ps.first && ps.final_) { LP_(false, true, 2); return; }
        if (ps.path == 3 && ps.first && !ps.final_) { LP_(true, false, 3); return; }
        if (ps.path == 4 && !ps.first && ps.final_) { LP_(false, true, 4); return; }
    }
#undef LP_
#define L_(A, B, C) do { auto k = &tile_fft_kernel<T, A, B, C, MT, 0>; XRFT_LAUNCH(k, g, b, ps.lds, st, ps.g, ps.pr, ps.ep); } while (0)
    const int sel = (ps.first ? 4 : 0) | (ps.final_ ? 2 : 0) | (ps.generic ? 1 : 0);
    switch (sel) {
        case 0: L_(false, false, false); break;
        case 1: L_(false, false, true); break;
        case 2: L_(false, true, false); break;
        case 3: L_(false, true, true); break;
        case 4: L_(true, false, false); break;
        case 5: L_(true, false, true); break;
        case 6: L_(true, true, false); break;
        default: L_(true, true, true); break;
    }
#undef L_
}

void appendf(std::string& s, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    s += buf;
}

void describe_passes(std::string& s, const std::vector<Pass>& v, const char* name) {
    for (const Pass& p : v) {
        appendf(s, "  [%s] %-16s n=%d radix=", name, p.label.c_str(), p.g.n);
        for (int i = 0; i < p.g.nr; ++i) appendf(s, "%s%d", i ? "x" : "", p.g.radix[i]);
        appendf(s, " T=%d threads=%d lds=%zuB%s%s%s%s\n", p.g.T, p.threads, p.lds, p.g.r2c ? " r2c" : "",
                p.first ? " first" : "", p.final_ ? " final" : "", p.generic ? " generic-radix" : "");
    }
}

}  // namespace

int upload_real_table(xrfthip_plan* P, DevBuf& buf, const double* h, int64_t n, int cplx) {
    if (!h) { buf.clear(); return XRFTHIP_OK; }
    const size_t cnt = (size_t)n * (cplx ? 2 : 1);
    if (P->dbl) return buf.upload(h, cnt * sizeof(double));
    std::vector<float> f(cnt);
    for (size_t i = 0; i < cnt; ++i) f[i] = (float)h[i];
    return buf.upload(f.data(), cnt * sizeof(float));
}


// workgroups per slab of radial_binsum_det_kernel: chunks of <= 2^17 elements (its int64 sums hold 2^17 values), at most 128
int iso_chunk_count(long long total) {
    long long c = std::max<long long>(1, std::min<long long>(128, total / 16384));
    while ((total + c - 1) / c > (1LL << 17)) ++c;
    return (int)c;
}
// bins per launch: the int64 sums and the exponent table of a window of bins share 64 KB of LDS
int iso_bin_window(bool cplx) { return (int)((64 * 1024) / (cplx ? 24 : 16)); }  // (+ 4 bytes per bin: the non-finite flags)

// radial sums of `bc` stored spectra [bc][ny][nxo] (rows / columns rotated by sy / sx) -> iso[bc][nbins (x2)], bit-reproducible
int run_radial_sums(int32_t dtype, const void* spec, const int32_t* d_binmap, long long bc, long long ny, long long nxo, int sy, int sx,
                           int nbins, int chunks, double* part, double* iso, hipStream_t st) {
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const int hw = cplx ? 2 : 1, win = iso_bin_window(cplx);
    const long long total = ny * nxo;
    const size_t esz = (dbl ? 8 : 4) * (size_t)hw;
    for (long long s0 = 0; s0 < bc; s0 += 32768) {  // grid.y limit
        const long long sc = std::min<long long>(32768, bc - s0);
        const void* src = (const char*)spec + (size_t)s0 * total * esz;
        double* pdst = part + (size_t)s0 * chunks * nbins * hw;
        for (int b0 = 0; b0 < nbins; b0 += win) {
            const int nb = std::min(win, nbins - b0);
            const dim3 grid((unsigned)chunks, (unsigned)sc), block(256);
            int ncopy = 1;  // copies of the tables (lanes spread over them: neighbouring samples share bins), as many as fit 32 KB
            while (ncopy < 8 && (size_t)nb * (cplx ? 20 : 12) * (2 * ncopy) <= 32 * 1024) ncopy *= 2;
            const size_t lds = (size_t)nb * (cplx ? 20 : 12) * ncopy + (size_t)nb * 4;
#define ISO_(TT, CC) do { auto k = &radial_binsum_det_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, lds, st, src, (const int*)d_binmap, total, (int)nxo, (int)ny, sy, sx, b0, nb, nbins, ncopy, pdst); } while (0)
            if (dbl) { if (cplx) ISO_(double, true); else ISO_(double, false); } else { if (cplx) ISO_(float, true); else ISO_(float, false); }
#undef ISO_
        }
        auto kr = &iso_reduce_kernel;
        XRFT_LAUNCH(kr, dim3((unsigned)((nbins * hw + 63) / 64), (unsigned)sc), dim3(256), 4 * 64 * sizeof(double), st, (const double*)pdst,
                    iso + (size_t)s0 * nbins * hw, chunks, nbins * hw, (const unsigned*)nullptr, hw);
    }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

void layout_workspace(xrfthip_plan* P) {
    const xrfthip_desc& d = P->d;
    if (P->fastr || P->fasts || P->fastg || P->fastgy) { P->G = (int)std::max<long long>(1, std::min<long long>(d.batch, 1 << 30)); P->ws_bytes = 0; return; }  // one pass, registers + LDS: no intermediate
    if (P->fastyc) {  // the tiled intermediate of one group of slabs
        long long G = d.slabs_per_group > 0 ? d.slabs_per_group : (P->tune_fast_group > 0 ? P->tune_fast_group : std::max<long long>(1, (32LL * 4096 * 4096) / (d.ny * d.nx)));
        G = std::max<long long>(1, std::min<long long>(G, std::max<long long>(d.batch, 1)));
        P->G = (int)G;
        P->off_w = 0;
        size_t w2_cols = (size_t)d.nx;  // complex columns of the intermediate per row
        if (d.flags & XRFTHIP_C2R_X) { const size_t cw = 2 * (size_t)ycols_geom(d.ny).gxy; w2_cols = (size_t)d.nx / 2 + cw; }  // (+ the block of the Nyquist column)
        P->ws_bytes = (((size_t)G * (size_t)d.ny * w2_cols * sizeof(cf)) + 255) & ~(size_t)255;
        return;
    }
    const bool fast = fast_on(P);
    long long G = d.slabs_per_group > 0 ? d.slabs_per_group : P->tune_group;
    size_t slab_w = (size_t)d.ny * std::max(P->width, P->w_cols) * P->csize;
    const bool yf = fast && P->yfirst;
    if (fast) {
        slab_w = (size_t)P->y_nrow_pad * (size_t)(P->y_pitch > 0 ? P->y_pitch : P->ynx) * (P->fastm ? P->csize : sizeof(cf));
        if (G <= 0) G = P->tune_fast_group > 0 ? P->tune_fast_group : std::max<long long>(1, (64LL * 4096 * 4096) / (d.ny * d.nx));  // y-first, 4096^2: 16: 62.7, 32: 61.4 us per slab; 32 -> 64: 301-303 -> 306-307 GFFT/s (tails, launch gaps and the plane-fit bubble amortise)
    }
    if (G <= 0) {
        // the Infinity Cache adds no bandwidth, so groups are sized for launch efficiency, not residency
        const size_t target = (size_t)P->tune_group_bytes;
        G = (long long)std::max<size_t>(1, target / std::max<size_t>(slab_w, 1));
    }
    G = std::max<long long>(1, std::min<long long>(G, std::max<long long>(d.batch, 1)));
    P->G = (int)G;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int nf = (d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE) ? 2 : 1;
    size_t off = 0;
    const size_t ncoef = (size_t)d.batch * ((d.flags & XRFTHIP_AXIS_Y) ? (size_t)d.nx : 1);  // trend per slab, or per column
    P->mom_chunks = (int)std::max<long long>(1, std::min<long long>(d.ny, (2048 + G - 1) / G));
    P->off_acc = off; off = al(off + (size_t)G * P->mom_chunks * 6 * sizeof(double) * nf);  // per-chunk partial sums of ONE group of slabs
    P->off_coef = off; off = al(off + ncoef * 6 * sizeof(double) * nf);
    bool need_w = d.ndim == 2 || yf, need_w2 = false;
    for (const Pass& p : P->passes) { if (p.out_kind == B_W2) need_w2 = true; if (p.out_kind == B_W) need_w = true; }
    P->off_w = off; if (need_w) off = al(off + (size_t)G * slab_w * (yf ? nf : 1));  // (y first: field 1's intermediate follows field 0's)
    P->off_w2 = off; if (need_w2) off = al(off + (size_t)G * d.ny * d.nx * P->csize);
    P->off_f0 = off; if (nf == 2 && !fast) off = al(off + (size_t)G * slab_w);
    const size_t nfit = (size_t)(yf ? 2 * P->ynx : d.ny);  // per-column sums + subtracted lines
    P->off_rowfit = off; if (fast) off = al(off + (size_t)G * nfit * 2 * sizeof(double) * (yf ? nf : 1));
    P->off_corr = off; if (fast) off = al(off + (size_t)G * nfit * 2 * sizeof(float) * (yf ? nf : 1));  // (16 bytes per column: fasty uses 8, fastm's float64 pairs all 16)
    P->off_isopart = off;
    if (yf && !P->fastm && (d.flags & XRFTHIP_ISO)) {  // per-workgroup partial radial sums of one group of slabs (reduced in order)
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS;
        const long long gx = fasty_rows_gx(P);  // YRows<NX>::GX
        const size_t upr = (size_t)P->y_nrow_pad / (two ? gx : 2 * gx);
        off = al(off + (size_t)G * upr * P->nbins * (two ? 2 : 1) * sizeof(double));
    }
    P->off_rdv = off;
    if (yf && !P->fastm && ((P->tune_y >> 21) & 1)) off = al(off + (size_t)G * (size_t)std::max<long long>(P->ynx / 8, 1) * sizeof(unsigned));  // (tuning: rendezvous counters of pass 1)
    P->off_isotmp = off;
    if (fastm_iso_fused(P)) {  // fastm with the radial sums inside pass 2: one partial table per row workgroup
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS;
        P->off_isopart = off;
        off = al(off + (size_t)G * (P->y_nrow_pad / fastm_rows_rpu(P)) * P->nbins * (two ? 2 : 1) * sizeof(double));
    } else if ((!fast || P->fastm) && (d.flags & XRFTHIP_ISO)) {  // generic and fastm kernels: the spectrum is stored (into the caller's array, or here), then summed
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS;
        const size_t out_esz = two ? P->csize : P->rsize;
        const long long total = d.ny * P->nx_out;
        P->iso_chunks = iso_chunk_count(total);
        if (d.flags & XRFTHIP_NO_SPECTRUM_OUT) off = al(off + (size_t)G * total * out_esz);
        P->off_isopart = off;
        off = al(off + (size_t)G * P->iso_chunks * std::max(P->nbins, 1) * (two ? 2 : 1) * sizeof(double));
    }
    P->ws_bytes = off;
}

xrfthip_plan::ProfRec* prof_begin(const xrfthip_plan* P, const std::string& label, hipStream_t st) {
    if (!P->prof || P->prof_recs.size() + 1 >= P->prof_recs.capacity()) return nullptr;
    xrfthip_plan* M = const_cast<xrfthip_plan*>(P);
    xrfthip_plan::ProfRec r;
    r.label = label;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
    (void)hipEventRecord(r.a, st);
    M->prof_recs.push_back(r);
    return &M->prof_recs.back();
}
void prof_end(xrfthip_plan::ProfRec* r, hipStream_t st) {
    if (r) (void)hipEventRecord(r->b, st);
}

template <typename T>
static int run_moments(const xrfthip_plan* P, const void* in, long long g0, long long gc, double* acc, double* coef, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const long long total = d.ny * d.nx;
    const long long chunks = P->mom_chunks;
    const size_t esz = P->cplx_in ? P->csize : P->rsize;
    if (d.flags & XRFTHIP_AXIS_Y) {  // one line (or mean) per column, straight into the coefficient table
        xrfthip_plan::ProfRec* recc = prof_begin(P, "column_fit", st);
        for (long long b0 = 0; b0 < gc; b0 += 32768) {
            const long long bc = std::min<long long>(32768, gc - b0);
            const dim3 grid((unsigned)((d.nx + 63) / 64), (unsigned)bc), block(256);  // (64 columns x 4 row parts per workgroup)
            const void* src = (const char*)in + (size_t)(g0 + b0) * total * esz;
            double* cdst = coef + (g0 + b0) * d.nx * 6;
            if (P->cplx_in) { auto k = &column_fit_kernel<T, true>; XRFT_LAUNCH(k, grid, block, 4 * 4 * 64 * sizeof(double), st, src, (long long)d.ny, (long long)d.nx, cdst, (int)d.detrend); }
            else { auto k = &column_fit_kernel<T, false>; XRFT_LAUNCH(k, grid, block, 4 * 4 * 64 * sizeof(double), st, src, (long long)d.ny, (long long)d.nx, cdst, (int)d.detrend); }
        }
        prof_end(recc, st);
        HIP_TRY(hipGetLastError());
        return XRFTHIP_OK;
    }
    const size_t lds = 6 * 256 * sizeof(double);
    xrfthip_plan::ProfRec* rec = prof_begin(P, "moments", st);
    for (long long b0 = 0; b0 < gc; b0 += 32768) {  // grid.y is limited to 65535 blocks
        const long long bc = std::min<long long>(32768, gc - b0);
        const dim3 grid((unsigned)chunks, (unsigned)bc), block(256);
        const void* src = (const char*)in + (size_t)(g0 + b0) * total * esz;
        if (P->cplx_in) { auto k = &slab_moments_kernel<T, true>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)d.ny, (long long)d.nx, total, (long long)d.nx, acc + b0 * chunks * 6); }
        else { auto k = &slab_moments_kernel<T, false>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)d.ny, (long long)d.nx, total, (long long)d.nx, acc + b0 * chunks * 6); }
    }
    prof_end(rec, st);
    rec = prof_begin(P, "finalize_coef", st);
    auto kf = &finalize_coef_kernel;
    XRFT_LAUNCH(kf, dim3((unsigned)gc), dim3(64), 0, st, (const double*)acc, coef + g0 * 6, gc, (long long)d.ny, (long long)d.nx, (int)d.detrend, (int)chunks);
    prof_end(rec, st);
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}


// Everything xrfthip_exec needs beyond the caller's buffers is built HERE, when the plan is created or one of its tables is
// set: window spectra and phase tables of the specialised paths (device allocations + blocking copies) and the workspace
// layout.  xrfthip_exec itself takes the plan as const: no allocation, no copy, no synchronisation, no getenv.
int finalize_plan(xrfthip_plan* P) {
    if (P->fusedi) return fusedi_tables(P);  // (its workspace layout does not depend on the tables)
    // the radial sums of a cross spectrum with a true-phase factor that is not 1 (two fields with different lags) need the factor per sample: the other paths
    if (P->fastg && P->d.out_mode == XRFTHIP_OUT_CROSS && (P->d.flags & XRFTHIP_ISO) && phase_nontrivial(P)) P->fastg = false;
    if (P->fastg || P->fastgy) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fasts) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fastr || P->fastyc) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
        if (P->fastyc_fs) {
            // a window has no separable form over the view; an input phase (PHASE_IN: the lag's factor on the source samples) must be one -- exp(i theta n) is:
            // row factor ph[256 i1], column factor ph[i2] / ph[0]; checked, else the generic passes take the plan
            bool ok = P->host_win_x.empty() && !P->win[1].p;
            if (ok && (P->d.flags & XRFTHIP_PHASE_IN) && P->fph_on) {
                const std::vector<double>& h = P->host_phase[1];
                const long long n = P->d.nx, vy = n / 256;
                ok = (long long)h.size() >= 2 * n;
                std::vector<cf> py((size_t)vy), px(256);
                if (ok) {
                    const double r0 = h[0], i0 = h[1], m0 = r0 * r0 + i0 * i0;
                    for (long long i1 = 0; i1 < vy; ++i1) { py[(size_t)i1].re = (float)h[(size_t)(512 * i1)]; py[(size_t)i1].im = (float)h[(size_t)(512 * i1 + 1)]; }
                    for (int i2 = 0; i2 < 256; ++i2) {  // ph[i2] conj(ph[0]) / |ph[0]|^2
                        const double re = h[(size_t)(2 * i2)], im = h[(size_t)(2 * i2 + 1)];
                        px[(size_t)i2].re = (float)((re * r0 + im * i0) / m0); px[(size_t)i2].im = (float)((im * r0 - re * i0) / m0);
                    }
                    double worst = 0.0;
                    for (long long nn = 0; nn < n; nn += 97) {  // (a sample of the products)
                        const long long i1 = nn / 256; const int i2 = (int)(nn % 256);
                        const double yr = h[(size_t)(512 * i1)], yi = h[(size_t)(512 * i1 + 1)], xr = (h[(size_t)(2 * i2)] * r0 + h[(size_t)(2 * i2 + 1)] * i0) / m0, xi = (h[(size_t)(2 * i2 + 1)] * r0 - h[(size_t)(2 * i2)] * i0) / m0;
                        worst = std::max(worst, std::hypot(yr * xr - yi * xi - h[(size_t)(2 * nn)], yr * xi + yi * xr - h[(size_t)(2 * nn + 1)]));
                    }
                    ok = worst < 1e-9 && m0 > 0.0;
                }
                if (ok) {
                    int rc = P->fph[0].upload(py.data(), py.size() * sizeof(cf));
                    if (!rc) rc = P->fs_phx.upload(px.data(), px.size() * sizeof(cf));
                    if (rc) return rc;
                }
            }
            if (!ok) P->fastyc = P->fastyc_fs = false;  // (the generic four-step passes)
        }
    } else if (P->fastmx) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fastmy) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fastm) {
        int rc = fasty_window_spectra(P);
        if (!rc && P->d.out_mode != XRFTHIP_OUT_POWER) rc = fast_phase_tables(P);
        if (rc) return rc;
    } else if (P->fast1d) {
        P->fast1d_win = !P->host_win_x.empty();
        int rc = P->fast1d_win ? fasty_window_spectra_1d(P) : fasty_window_spectra(P);
        if (!rc && P->d.out_mode != XRFTHIP_OUT_POWER) rc = fast_phase_tables(P);
        if (rc) return rc;
    } else if (P->fast4096) {
        int rc = XRFTHIP_OK;
        if (P->yfirst) {
            rc = fasty_window_spectra(P);
            if (!rc && P->d.out_mode != XRFTHIP_OUT_POWER) rc = fast_phase_tables(P);
        }
        if (rc) return rc;
    }
    layout_workspace(P);
    return XRFTHIP_OK;
}

template <typename T>
static int run_pipeline(const xrfthip_plan* P, const std::vector<Pass>& passes, const void* in, void* out, double* iso,
                        char* ws, const double* coef, long long g0, long long gc, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const size_t in_esz = P->cplx_in ? P->csize : P->rsize;
    const void* iso_src = nullptr;
    for (const Pass& p0 : passes) {
        Pass p = p0;
        p.g.n_outer = p.outer_per_slab * gc;
        p.g.n_tiles = p.g.tile_axis == 0 ? (p.g.n_outer + p.g.T - 1) / p.g.T : p.g.n_outer * p.g.tiles_per_outer;
        auto buf = [&](int kind) -> void* {
            switch (kind) {
                case B_W: return ws + P->off_w;
                case B_W2: return ws + P->off_w2;
                case B_F0: return ws + P->off_f0;
                default: return nullptr;
            }
        };
        if (p.first) {
            p.pr.in = (const char*)in + (size_t)g0 * d.ny * ((d.flags & XRFTHIP_C2R_X) ? d.nx / 2 + 1 : d.nx) * in_esz;
            p.pr.win_y = P->win[0].p;
            p.pr.win_x = P->win[1].p;
            p.pr.coef = coef ? coef + g0 * ((d.flags & XRFTHIP_AXIS_Y) ? d.nx : 1) * 6 : nullptr;
            if (!coef) p.pr.detrend = 0;
            if (d.flags & XRFTHIP_PHASE_IN) { p.pr.ph_y = P->phase[0].p; p.pr.ph_x = P->phase[1].p; }
        } else {
            p.g.in = buf(p.in_kind);
        }
        if (p.final_) {
            if (p.out_kind == B_F0) {
                p.ep.out = buf(B_F0);
            } else {
                const size_t out_esz = (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_PHASE || (d.flags & XRFTHIP_C2R_X)) ? P->rsize : P->csize;
                p.ep.out = out ? (char*)out + (size_t)g0 * d.ny * P->nx_out * out_esz : nullptr;
                if ((d.flags & XRFTHIP_ISO) && iso && !out) p.ep.out = ws + P->off_isotmp;  // isotropic spectra: the full spectrum of this group lives in the workspace
                iso_src = p.ep.out;
                if (!(d.flags & XRFTHIP_PHASE_IN)) { p.ep.ph_y = P->phase[0].p; p.ep.ph_x = P->phase[1].p; }
                p.ep.other = (d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE) ? buf(B_F0) : nullptr;
            }
        } else {
            p.g.out = buf(p.out_kind);
        }
        if (p.g.n_tiles <= 0) continue;
        const int grid = (int)std::min<long long>(p.g.n_tiles, P->tune_max_grid);
        xrfthip_plan::ProfRec* rec = prof_begin(P, p.label, st);
        launch_tile<T>(p, grid, st);
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
    }
    if ((d.flags & XRFTHIP_ISO) && iso && iso_src) {  // radial sums of the stored spectrum (xrft.py:895-906), bit-reproducible
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS;
        const int32_t sdt = two ? (P->dbl ? XRFTHIP_C128 : XRFTHIP_C64) : (P->dbl ? XRFTHIP_F64 : XRFTHIP_F32);
        xrfthip_plan::ProfRec* rec = prof_begin(P, "radial_sums", st);
        const int rc = run_radial_sums(sdt, iso_src, (const int32_t*)P->binmap.p, gc, d.ny, P->nx_out, (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0,
                                       (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0, P->nbins, P->iso_chunks, reinterpret_cast<double*>(ws + P->off_isopart),
                                       iso + (size_t)g0 * P->nbins * (two ? 2 : 1), st);
        prof_end(rec, st);
        if (rc) return rc;
    }
    return XRFTHIP_OK;
}

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int xrfthip_version(void) { return XRFTHIP_VERSION; }

const char* xrfthip_strerror(int status) {
    switch (status) {
        case XRFTHIP_OK: return "ok";
        case XRFTHIP_BAD_ARG: return "bad argument";
        case XRFTHIP_UNSUPPORTED_LENGTH: return "unsupported transform length (prime factor > 128 or does not fit LDS)";
        case XRFTHIP_WORKSPACE_TOO_SMALL: return "workspace too small";
        case XRFTHIP_HIP_ERROR: return "HIP runtime error (see xrfthip_last_hip_error)";
        case XRFTHIP_ALLOC_FAILED: return "device allocation failed";
        case XRFTHIP_MISSING_TABLE: return "a required table (window / phase / bin map) was not set";
        default: return "unknown status";
    }
}

int xrfthip_last_hip_error(void) { return g_last_hip_error; }


int xrfthip_plan_create(xrfthip_plan** plan, const xrfthip_desc* desc) {
    // (a descriptor of the version before `inner` was appended is accepted: inner = 1)
    constexpr uint32_t kOldDescSize = (uint32_t)offsetof(xrfthip_desc, inner), kOldDescSize2 = (uint32_t)offsetof(xrfthip_desc, mid);
    if (!plan || !desc || (desc->struct_size != sizeof(xrfthip_desc) && desc->struct_size != kOldDescSize && desc->struct_size != kOldDescSize2)) return XRFTHIP_BAD_ARG;
    xrfthip_desc dcopy{};
    memcpy(&dcopy, desc, desc->struct_size);
    dcopy.struct_size = sizeof(xrfthip_desc);
    if (dcopy.inner < 0 || dcopy.mid < 0) return XRFTHIP_BAD_ARG;
    if (dcopy.inner == 0) dcopy.inner = 1;
    if (dcopy.mid == 0) dcopy.mid = 1;
    const xrfthip_desc& d = dcopy;
    if (d.ndim != 1 && d.ndim != 2) return XRFTHIP_BAD_ARG;
    if (d.batch < 0 || d.nx < 1 || d.ny < 1 || (d.ndim == 1 && d.ny != 1)) return XRFTHIP_BAD_ARG;
    if (d.nx > (1LL << 30) || d.ny > (1LL << 30) || d.nx * d.ny > (1LL << 31) - 1) return XRFTHIP_BAD_ARG;  // per-element index math is 32-bit
    if (d.dtype < XRFTHIP_F32 || d.dtype > XRFTHIP_C128) return XRFTHIP_BAD_ARG;
    if (d.out_mode < XRFTHIP_OUT_COMPLEX || d.out_mode > XRFTHIP_OUT_PHASE) return XRFTHIP_BAD_ARG;
    if ((d.flags & (XRFTHIP_INVERSE | XRFTHIP_C2R_X | XRFTHIP_PHASE_IN)) && (d.dtype < XRFTHIP_C64 || d.out_mode != XRFTHIP_OUT_COMPLEX || d.detrend)) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_C2R_X) && (!(d.flags & XRFTHIP_INVERSE) || (d.nx & 1) || (d.flags & (XRFTHIP_ISHIFT_X | XRFTHIP_FLIP_X)))) return XRFTHIP_BAD_ARG;
    if (d.detrend < XRFTHIP_DETREND_NONE || d.detrend > XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    const bool cplx_in = d.dtype >= XRFTHIP_C64;
    if ((d.flags & XRFTHIP_HALF_X) && cplx_in) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_HALF_X) && (d.flags & (XRFTHIP_SHIFT_X | XRFTHIP_SHIFT_Y))) return XRFTHIP_BAD_ARG;  // xrft.py:403
    if ((d.flags & XRFTHIP_HALF_Y) && (cplx_in || !(d.inner > 1 || d.mid > 1) || (d.ny & 1) || (d.flags & (XRFTHIP_HALF_X | XRFTHIP_SHIFT_X | XRFTHIP_SHIFT_Y | XRFTHIP_AXIS_Y)))) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_REALDIM_X2) && (!(d.flags & (XRFTHIP_HALF_X | XRFTHIP_HALF_Y)) || d.out_mode == XRFTHIP_OUT_COMPLEX)) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_ISO) && (d.ndim != 2 || d.out_mode == XRFTHIP_OUT_COMPLEX)) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_NO_SPECTRUM_OUT) && !(d.flags & XRFTHIP_ISO)) return XRFTHIP_BAD_ARG;
    if (d.ndim == 1 && (d.flags & (XRFTHIP_SHIFT_Y | XRFTHIP_ISHIFT_Y | XRFTHIP_FLIP_Y))) return XRFTHIP_BAD_ARG;
    if ((d.flags & (XRFTHIP_FLIP0_Y | XRFTHIP_FLIP0_X)) && d.out_mode != XRFTHIP_OUT_CROSS && d.out_mode != XRFTHIP_OUT_PHASE) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_FLIP0_Y) && d.ndim == 1) return XRFTHIP_BAD_ARG;
    if ((d.flags & XRFTHIP_AXIS_Y) && (d.ndim != 2 || (d.flags & XRFTHIP_FLIP0_X) || (d.flags & (XRFTHIP_SHIFT_X | XRFTHIP_ISHIFT_X | XRFTHIP_FLIP_X |
                                                                    XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT | XRFTHIP_C2R_X)))) return XRFTHIP_BAD_ARG;  // (PHASE_IN: only where fastgy takes the plan, below)
    // AXIS_Y with HALF_X / REALDIM_X2 (ABI 0.1.4): real_dim along the ONE transformed axis -- ny / 2 + 1 rows per slab, unshifted; the one-pass kernels only (below)
    if ((d.flags & XRFTHIP_AXIS_Y) && (d.flags & (XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2)) &&
        (cplx_in || !(d.flags & XRFTHIP_HALF_X) || (d.flags & (XRFTHIP_SHIFT_Y | XRFTHIP_FLIP_Y | XRFTHIP_INVERSE)))) return XRFTHIP_BAD_ARG;

    if (d.inner > 1 || d.mid > 1) return create_inner_plan(plan, d);

    xrfthip_plan* P = new (std::nothrow) xrfthip_plan();
    if (!P) return XRFTHIP_ALLOC_FAILED;
    P->d = d;
    P->tune_group = env_ll("XRFTHIP_GROUP", 0);
    P->tune_fast_group = env_ll("XRFTHIP_FAST_GROUP", 0);
    P->tune_y = env_ll("XRFTHIP_YTUNE", kYTuneDefault);
    P->tune_isorows = env_ll("XRFTHIP_ISOROWS", 0);
    P->tune_group_bytes = env_ll("XRFTHIP_GROUP_BYTES", 512LL << 20);
    P->tune_cols_grid = env_ll("XRFTHIP_FAST_COLS_GRID", kCUs);
    P->tune_max_grid = env_ll("XRFTHIP_MAX_GRID", 8 * kCUs * 4);
    P->cplx_in = cplx_in;
    P->dbl = d.dtype == XRFTHIP_F64 || d.dtype == XRFTHIP_C128;
    P->rsize = P->dbl ? 8 : 4;
    P->csize = 2 * P->rsize;
    P->nxh = cplx_in ? d.nx : d.nx / 2 + 1;
    P->nx_out = ((d.flags & XRFTHIP_HALF_X) && !(d.flags & XRFTHIP_AXIS_Y)) ? d.nx / 2 + 1 : d.nx;  // (AXIS_Y: the half is along y)
    // width of the intermediate: the half spectrum for real input, unless the row does not fit one LDS tile
    // (four-step along x computes every kx) -- decided inside build_x through P->width.
    P->width = P->nxh;
    if (d.flags & XRFTHIP_AXIS_Y) P->width = d.nx;  // x is not transformed: every column is its own sequence
    else if (!cplx_in) {
        const long long n_try = (d.nx % 2 == 0 && d.nx >= 2) ? d.nx / 2 : d.nx;
        TileChoice c = choose_tile(n_try, P->csize, false, 1LL << 40, 0);
        if (c.T == 0 || n_try >= env_ll("XRFTHIP_X_FOURSTEP_MIN", 1LL << 40)) P->width = d.nx;
    }
    P->mirror = !cplx_in && !(d.flags & (XRFTHIP_HALF_X | XRFTHIP_AXIS_Y)) && P->width == d.nx / 2 + 1 && d.nx > 1;
    auto fast_len = [](long long n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; };
    {
        const uint32_t shifts = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X, ish = XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X, isof = XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT;
        const uint32_t halff = XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2;  // real_dim: half output, no mirror
        const uint32_t allowed = d.out_mode == XRFTHIP_OUT_POWER ? (shifts | isof | halff) : d.out_mode == XRFTHIP_OUT_COMPLEX ? (shifts | ish | XRFTHIP_HALF_X)
                                 : d.out_mode == XRFTHIP_OUT_CROSS ? (shifts | ish | isof | halff) : d.out_mode == XRFTHIP_OUT_PHASE ? (shifts | ish | XRFTHIP_HALF_X) : 0u;
        P->fast4096 = d.ndim == 2 && fast_len(d.ny) && fast_len(d.nx) && d.dtype == XRFTHIP_F32 &&
                      !(d.flags & ~allowed) && !((d.flags & halff) && (d.flags & XRFTHIP_ISO)) && !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X)) &&
                      !env_ll("XRFTHIP_NO_FAST", 0);
    }
    {   // complex float32 slabs of these lengths: the two-pass pipeline's complex form (fasty_c2c.h)
        const uint32_t okc = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X | XRFTHIP_INVERSE | XRFTHIP_PHASE_IN | XRFTHIP_C2R_X) : 0u);
        const bool c2r = (d.flags & XRFTHIP_C2R_X) != 0;  // (irfftn: the half spectrum in, nx real samples per row out; the row transforms have nx/2 points)
        P->fastyc = d.ndim == 2 && d.dtype == XRFTHIP_C64 && fast_len(d.ny) && (c2r ? (d.nx % 2 == 0 && fast_len(d.nx / 2)) : fast_len(d.nx)) && !d.detrend &&
                    (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) && !(d.flags & ~okc) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTYC", 1) != 0;
        if (P->fastyc) {
            std::vector<float> ones((size_t)4096, 1.0f);
            int rcc = build_twiddle<float>(P->tw_fx, c2r ? d.nx / 2 : d.nx, c2r ? d.nx / 2 : d.nx);
            if (!rcc && c2r) rcc = build_twiddle<float>(P->tw_big1d, d.nx, d.nx / 32);  // W_nx^u, u < (nx/2) / 16
            if (!rcc) rcc = build_twiddle<float>(P->tw_fy, d.ny, d.ny);
            if (!rcc) rcc = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rcc) { delete P; return rcc; }
            P->yny = d.ny; P->ynx = d.nx;
        }
    }
    {   // ONE long complex float32 sequence per batch entry, 2^16 .. 2^20 points (xrft.ifft of the spectrum of a long row, fft of complex rows: the inverse twin of
        // BASELINE config 2): the same two passes on the [n / 256][256] view, pass 2 in its four-step form (before: the generic four-step passes, 49 GFFT/s)
        const uint32_t okf = XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_X | XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u);
        const bool fs = d.ndim == 1 && d.dtype == XRFTHIP_C64 && d.nx >= 65536 && d.nx <= 1048576 && (d.nx & (d.nx - 1)) == 0 && !d.detrend &&
                        (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) && !(d.flags & ~okf) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTYC", 1) != 0;
        if (fs) {
            P->fastyc = P->fastyc_fs = true;
            std::vector<float> ones((size_t)4096, 1.0f);
            int rcc = build_twiddle<float>(P->tw_fx, 256, 256);
            if (!rcc) rcc = build_twiddle<float>(P->tw_big1d, d.nx, d.nx / 16);  // W_N^j, j < N / 16: the four-step twiddles of a row (k1 u, k1 NT)
            if (!rcc) rcc = build_twiddle<float>(P->tw_fy, d.nx / 256, d.nx / 256);
            if (!rcc) rcc = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rcc) { delete P; return rcc; }
            P->yny = d.nx / 256; P->ynx = 256;
        }
    }
    // a small float32 slab (64 | 128 | 256 points per axis) fits the registers of one workgroup: full power spectra in ONE pass (fasts.h)
    {
        auto small_len = [](long long n) { return n == 64 || n == 128 || n == 256; };
        const uint32_t oks = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_POWER ? (XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT) : (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X));
        P->fasts = d.ndim == 2 && d.dtype == XRFTHIP_F32 && small_len(d.ny) && small_len(d.nx) && (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_COMPLEX) &&
                   !(d.flags & ~oks) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTS", 1) != 0;
    }
    if (P->fasts) {  // (takes precedence over the two-pass pipeline wherever the plan is looked at; an isotropic plan whose bin map turns out
                     // not to be a radial one falls back to it: xrfthip_plan_set_binmap)
        P->tune_sgrid = env_ll("XRFTHIP_FASTS_GRID", -1);
        P->tune_sstagger = env_ll("XRFTHIP_FASTS_STAGGER", (3 << 8) | 2);  // (three classes 6.8 us apart: (4096, 256, 256) linear + Hann 310 -> 320 (the walk) -> 328 GFFT/s, profiles/r06_fasts_prefetch.txt)
        std::vector<float> ones((size_t)256, 1.0f);
        int rcs = build_twiddle<float>(P->tw_sy, d.ny, d.ny);
        if (!rcs) rcs = build_twiddle<float>(P->tw_sx, d.nx, d.nx);
        if (!rcs) rcs = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
        if (rcs) { delete P; return rcs; }
    }
    if (P->fast4096) {
        // every mode of these slabs takes the two-pass y-first pipeline (fasty.h)
        P->yfirst = true;
        if (P->yfirst) {
            P->yny = d.ny; P->ynx = d.nx;
            const int rpu = yrows_geom(d.nx).rk;
            P->y_nrow_pad = (int)((d.ny / 2 + 1 + rpu - 1) / rpu * rpu);
        }
        int rc4 = build_twiddle<float>(P->tw_fx, d.nx, d.nx);
        if (!rc4) rc4 = build_twiddle<float>(P->tw_fy, d.ny, d.ny);
        std::vector<float> ones((size_t)std::max(d.ny, d.nx), 1.0f);
        if (!rc4) rc4 = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
        if (rc4) { delete P; return rc4; }
    }
    {   // one real float32 row of 65536 samples per workgroup, transformed in registers in ONE pass (fastr.h): 12 bytes per sample through
        // memory where the four-step form below moves 28
        const uint32_t okr = XRFTHIP_SHIFT_X | XRFTHIP_HALF_X | (d.out_mode == XRFTHIP_OUT_POWER ? XRFTHIP_REALDIM_X2 : 0u) | (d.out_mode == XRFTHIP_OUT_COMPLEX ? XRFTHIP_ISHIFT_X : 0u);
        P->fastr = d.ndim == 1 && (d.nx == 65536 || d.nx == 32768 || d.nx == 16384 || d.nx == 8192 || d.nx == 4096) && d.dtype == XRFTHIP_F32 && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) &&
                   !(d.flags & ~okr) && !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X)) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTR", 1) != 0;
        // ... and complex rows of 2048 .. 16384 points (xrft.ifft / fft of complex data along the contiguous axis): the same transform without the packing and the split
        const uint32_t okc = XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_X | XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u);
        P->fastr_cin = !P->fastr && d.ndim == 1 && d.dtype == XRFTHIP_C64 && (d.nx == 16384 || d.nx == 8192 || d.nx == 4096 || d.nx == 2048) && !d.detrend &&
                       (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) && !(d.flags & ~okc) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTC", 1) != 0;
        // rows of 256 .. 4096 points: two rows per thread through one LDS buffer (the row pass of fasty_c2c.h on the input's own rows; XRFTHIP_CROWS=0: fastc_kernel / fastm_xonly_kernel)
        const bool c2r1 = (d.flags & XRFTHIP_C2R_X) != 0;  // (irfft along the contiguous axis: rows of nx/2 + 1 complex values in, nx real samples out)
        P->fastr_rows = !P->fastr && d.ndim == 1 && d.dtype == XRFTHIP_C64 && (c2r1 ? (d.nx % 2 == 0 && fast_len(d.nx / 2) && d.out_mode == XRFTHIP_OUT_COMPLEX) : fast_len(d.nx)) && !d.detrend &&
                        (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) && !(d.flags & ~(okc | (d.out_mode == XRFTHIP_OUT_COMPLEX ? XRFTHIP_C2R_X : 0u))) &&
                        !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_CROWS", 1) != 0;
        if (P->fastr_rows) {
            P->fastr = true;
            P->fastr_cin = false;
            std::vector<float> ones((size_t)4096, 1.0f);
            int rcr = build_twiddle<float>(P->tw_fx, c2r1 ? d.nx / 2 : d.nx, c2r1 ? d.nx / 2 : d.nx);
            if (!rcr && c2r1) rcr = build_twiddle<float>(P->tw_big1d, d.nx, d.nx / 32);
            if (!rcr) rcr = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rcr) { delete P; return rcr; }
        } else
        if (P->fastr_cin) {
            P->fastr = true;
            P->tune_rgrid = env_ll("XRFTHIP_FASTR_GRID", 0);
            P->tune_rstagger = env_ll("XRFTHIP_FASTR_STAGGER", 0);
            const long long thr = d.nx / 32;  // threads per row: 32 complex values each
            int rcr = build_twiddle<float>(P->tw_rm, d.nx, thr);
            if (!rcr) rcr = build_twiddle<float>(P->tw_rs, thr, 32);
            if (rcr) { delete P; return rcr; }
        } else
        if (P->fastr) {
            // 65536 samples: one resident workgroup per CU walks the rows (measured: 359 vs 344 GFFT/s for a workgroup per row, profiles/r04_fastr.txt);
            // the shorter rows (several workgroups per CU): a workgroup per row
            // 32768 / 16384 samples (one / two workgroups per CU): a resident set, too -- with the start stagger run_fastr picks (profiles/r06_rows_stagger.txt)
            P->tune_rgrid = env_ll("XRFTHIP_FASTR_GRID", d.nx == 65536 ? kCUs : (d.nx == 32768 && d.batch >= 2 * kCUs) ? kCUs : (d.nx == 16384 && d.out_mode == XRFTHIP_OUT_COMPLEX && d.batch >= 4 * kCUs) ? 2 * kCUs : 0);
            // two classes of workgroups 10 us apart: dft (1024, 65536) 388 -> 441 GFFT/s, power_spectrum 517 -> 586 (profiles/r06_c2_stagger.txt); -1: run_fastr's rule
            P->tune_rstagger = env_ll("XRFTHIP_FASTR_STAGGER", d.nx == 65536 ? ((2 << 8) | 3) : -1);
            const long long thr = d.nx / 64;  // threads per row: 32 packed complex values each
            int rcr = build_twiddle<float>(P->tw_rm, d.nx / 2, thr);
            if (!rcr) rcr = build_twiddle<float>(P->tw_rs, thr, 32);
            if (!rcr) rcr = build_twiddle<float>(P->tw_rn, d.nx, thr);
            if (rcr) { delete P; return rcr; }
        }
    }
    {   // one long real float32 sequence per slab, N = n1 * 256 samples (2^16 .. 2^20): the two passes of the y-first pipeline are
        // the two steps of its four-step transform (fasty.h, FS)
        const long long n1 = d.nx / 256;
        const bool pow2 = d.nx >= 65536 && d.nx <= (1LL << 20) && (d.nx & (d.nx - 1)) == 0;
        const uint32_t ok1 = XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? XRFTHIP_ISHIFT_X : 0u);
        P->fast1d = !P->fastr && d.ndim == 1 && pow2 && d.dtype == XRFTHIP_F32 && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) &&
                    !(d.flags & ~ok1) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FAST1D", 1) != 0;
        if (P->fast1d) {
            P->yfirst = true;
            P->yny = n1; P->ynx = 256;
            const int rpu = yrows_geom(256, true).rk;
            P->y_nrow_pad = (int)((n1 / 2 + 1 + rpu - 1) / rpu * rpu);
            int rc1 = build_twiddle<float>(P->tw_fx, 256, 256);
            if (!rc1) rc1 = build_twiddle<float>(P->tw_fy, n1, n1);
            if (!rc1) rc1 = build_twiddle<float>(P->tw_big1d, d.nx, d.nx / 2 + 1);
            std::vector<float> ones((size_t)std::max<long long>(n1, 256), 1.0f);
            if (!rc1) rc1 = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rc1) { delete P; return rc1; }
        }
    }
    {   // real float64 slabs on the regular lat/lon lengths: the mixed-radix form of the y-first pipeline (fastm.h)
        const uint32_t shifts = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X, ish = XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X;
        const uint32_t isof = XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT;  // radial sums: fused into pass 2, or a pass over the stored spectrum (run_radial_sums)
        const uint32_t halff = XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2;     // real_dim: half output, no mirror columns
        const uint32_t allowed = d.out_mode == XRFTHIP_OUT_POWER ? (shifts | isof | halff) : d.out_mode == XRFTHIP_OUT_CROSS ? (shifts | ish | isof | halff)
                                 : (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_PHASE) ? (shifts | ish | XRFTHIP_HALF_X) : 0u;
        const bool half_ok = !((d.flags & halff) && (d.flags & XRFTHIP_ISO)) && !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X));
        P->fastm = half_ok && d.ndim == 2 && (d.dtype == XRFTHIP_F64 || d.dtype == XRFTHIP_F32) && !P->fast4096 && fastm_len(d.ny, P->dbl) && fastm_len(d.nx, P->dbl) && !(d.flags & ~allowed) &&
                   !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTM", 1) != 0 && env_ll("XRFTHIP_FASTN_TABLES", 1) != 0;
        if (P->fastm) {
            const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
            const int rpu = fastm_rpu(d.nx, two, P->dbl);
            if (rpu < 1 || rpu % fastm_rk2(d.ny, d.nx, two, P->dbl) != 0 || d.nx % fastm_cw(d.ny, d.nx, P->dbl) != 0) P->fastm = false;
        }
        if (P->fastm) {
            const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
            const int rpu = two ? fastm_rpu(d.nx, true, P->dbl) : mgeom(d.nx, P->dbl).g;  // (the largest count a row kernel of this plan may use)
            P->yfirst = true;
            P->yny = d.ny; P->ynx = d.nx; P->y_pitch = d.nx;
            P->y_nrow_pad = (int)((d.ny / 2 + 1 + rpu - 1) / rpu * rpu);
            int rcm = P->dbl ? build_twiddle<double>(P->tw_fx, d.nx, d.nx) : build_twiddle<float>(P->tw_fx, d.nx, d.nx);
            if (!rcm) rcm = P->dbl ? build_twiddle<double>(P->tw_fy, d.ny, d.ny) : build_twiddle<float>(P->tw_fy, d.ny, d.ny);
            std::vector<double> ones((size_t)std::max(d.ny, d.nx), 1.0);
            std::vector<float> onesf((size_t)std::max(d.ny, d.nx), 1.0f);
            if (!rcm) rcm = P->dbl ? P->ones4096.upload(ones.data(), ones.size() * sizeof(double)) : P->ones4096.upload(onesf.data(), onesf.size() * sizeof(float));
            if (rcm) { delete P; return rcm; }
        }
    }
    {   // one transform axis that is not the contiguous one, real input: pass 1 of the same kernels is the whole transform
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
        const uint32_t allowed = XRFTHIP_AXIS_Y | XRFTHIP_SHIFT_Y | (d.out_mode != XRFTHIP_OUT_POWER ? XRFTHIP_ISHIFT_Y : 0u) |
                                 ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u) |  // (xrft.ifft along the axis)
                                 (!cplx_in ? (XRFTHIP_HALF_X | (d.out_mode != XRFTHIP_OUT_PHASE ? XRFTHIP_REALDIM_X2 : 0u)) : 0u);  // (real_dim along the axis: half output)
        P->fastmy = (d.flags & XRFTHIP_AXIS_Y) && d.ndim == 2 && (!cplx_in || !two) &&
                    (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER || two) && !(d.flags & ~allowed) && fastmy_len(d.ny, P->dbl) &&
                    !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_Y)) &&  // (the half output is unshifted: also refused by xrfthip_plan_create, kept here so the two cannot drift apart)
                    d.batch * d.nx < (1LL << 30) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTM", 1) != 0;
        if (P->fastmy && !(cplx_in && !two) && d.nx % ((two ? 1 : 2) * mygeom(d.ny, P->dbl).g) != 0) P->fastmy = false;  // (complex columns: any count, the last block guarded)
        if (P->fastmy) {
            int rcm = P->dbl ? build_twiddle<double>(P->tw_fy, d.ny, d.ny) : build_twiddle<float>(P->tw_fy, d.ny, d.ny);
            std::vector<double> ones((size_t)d.ny, 1.0);
            std::vector<float> onesf((size_t)d.ny, 1.0f);
            if (!rcm) rcm = P->dbl ? P->ones4096.upload(ones.data(), ones.size() * sizeof(double)) : P->ones4096.upload(onesf.data(), onesf.size() * sizeof(float));
            if (rcm) { delete P; return rcm; }
        }
    }
    {   // ... on any other smooth length: one pass in LDS with the radices as data (fastg.h: fastgy_kernel)
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;  // (two REAL fields: a column of each = one packed sequence; no flipped field)
        const uint32_t allowed = XRFTHIP_AXIS_Y | XRFTHIP_SHIFT_Y | (d.out_mode != XRFTHIP_OUT_POWER ? XRFTHIP_ISHIFT_Y : 0u) |
                                 ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u) |  // (xrft.ifft along the axis: conj in, conj out, the input rotated)
                                 (!cplx_in ? (XRFTHIP_HALF_X | (d.out_mode != XRFTHIP_OUT_PHASE ? XRFTHIP_REALDIM_X2 : 0u)) : 0u);
        P->fastgy = !P->fastmy && (d.flags & XRFTHIP_AXIS_Y) && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER || (two && !cplx_in)) && !(d.flags & ~allowed) &&
                    !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_Y)) &&
                    !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTG", 1) != 0 && fastgy_try(P);
        if (P->fastgy) {
            const long long m = P->gy_blue_m ? P->gy_blue_m : d.ny;  // length of the passes
            int rcg = P->dbl ? build_twiddle<double>(P->g_twy, m, m) : build_twiddle<float>(P->g_twy, m, m);
            if (!rcg && !P->gy_rad_p) rcg = fastg_rev(P->g_ry, (int)m, P->g_revy, P->g_hrevy);
            if (!rcg && P->gy_rad_p) rcg = P->dbl ? fastgy_rader_tables<double>(P) : fastgy_rader_tables<float>(P);
            if (!rcg && P->gy_blue_m) rcg = P->dbl ? fastgy_blue_tables<double>(P) : fastgy_blue_tables<float>(P);
            if (rcg) { delete P; return rcg; }
        }
    }
    {   // ... and along the CONTIGUOUS axis of a 1-D plan when the length holds ONE prime 17 ... 127 (365 / 730 / 1460-sample (station, time) rows): the same kernel's
        // prime-factor / Rader form with the lanes along the samples (fastg.h, FORM 3); every other 1-D length has its kernels below
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
        const uint32_t allowed = XRFTHIP_SHIFT_X | (d.out_mode != XRFTHIP_OUT_POWER ? XRFTHIP_ISHIFT_X : 0u) |
                                 ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u) |
                                 (!cplx_in ? (XRFTHIP_HALF_X | (d.out_mode != XRFTHIP_OUT_PHASE ? XRFTHIP_REALDIM_X2 : 0u)) : 0u);
        const bool half_ok = !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X)) && !((d.flags & XRFTHIP_REALDIM_X2) && !(d.flags & XRFTHIP_HALF_X));
        if (half_ok && !P->fastgy && d.ndim == 1 && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER || (two && !cplx_in)) && !(d.flags & ~allowed) &&
            !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTG", 1) != 0 && fastgy_try(P, true)) {
            P->fastgy = true;
            int rcg = P->dbl ? build_twiddle<double>(P->g_twy, d.nx, d.nx) : build_twiddle<float>(P->g_twy, d.nx, d.nx);
            if (!rcg) rcg = P->dbl ? fastgy_rader_tables<double>(P) : fastgy_rader_tables<float>(P);
            if (rcg) { delete P; return rcg; }
        }
    }
    {   // one short transform axis, the contiguous one, real input: rows packed in pairs through the same three passes
        const bool two = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
        const uint32_t allowed = XRFTHIP_SHIFT_X | XRFTHIP_HALF_X | (d.out_mode != XRFTHIP_OUT_PHASE ? XRFTHIP_REALDIM_X2 : 0u) | (d.out_mode != XRFTHIP_OUT_POWER ? XRFTHIP_ISHIFT_X : 0u) |
                                 ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN | XRFTHIP_C2R_X) : 0u);  // (xrft.ifft along the contiguous axis: conj in, conj out, the input rotated; irfft: two half rows per transform)
        P->fastmx = !P->fastr && d.ndim == 1 && (!cplx_in || (!two && !(d.flags & (XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2)))) && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER || two) &&
                    !(d.flags & ~allowed) && !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X)) && !((d.flags & XRFTHIP_REALDIM_X2) && !(d.flags & XRFTHIP_HALF_X)) &&
                    fastmx_len(d.nx, P->dbl) && d.batch < (1LL << 31) - 16 && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTM", 1) != 0;
        if (P->fastmx) {
            int rcm = P->dbl ? build_twiddle<double>(P->tw_fx, d.nx, d.nx) : build_twiddle<float>(P->tw_fx, d.nx, d.nx);
            std::vector<double> ones((size_t)d.nx, 1.0);
            std::vector<float> onesf((size_t)d.nx, 1.0f);
            if (!rcm) rcm = P->dbl ? P->ones4096.upload(ones.data(), ones.size() * sizeof(double)) : P->ones4096.upload(onesf.data(), onesf.size() * sizeof(float));
            if (rcm) { delete P; return rcm; }
        }
    }
    {   // a small slab of any smooth shape, either precision, that none of the specialised kernels above takes: one pass in LDS (fastg.h)
        // complex input (fft of complex data, every inverse transform): power / complex, no detrend, no real_dim, no radial sums
        const bool cin_ok = !cplx_in || ((d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_COMPLEX) && !d.detrend && !(d.flags & (XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2 | XRFTHIP_ISO)) &&
                                         (!(d.flags & XRFTHIP_C2R_X) || !(d.nx & 1)));
        const uint32_t okg = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | XRFTHIP_HALF_X | ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN | XRFTHIP_C2R_X) : 0u) |
                             (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X)
                              : d.out_mode == XRFTHIP_OUT_CROSS ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X | XRFTHIP_REALDIM_X2 | XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT)  // (no flipped field: the other paths)
                              : (XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT | XRFTHIP_REALDIM_X2));
        // (a 1-D transform along x that neither the register kernels nor the table lengths take: the same kernel on groups of rows)
        const bool one_ok = d.ndim != 1 || (!P->fastr && !P->fastmx && !P->fast1d && !(d.flags & (XRFTHIP_ISO | XRFTHIP_SHIFT_Y | XRFTHIP_ISHIFT_Y)));
        P->fastg = one_ok && cin_ok && !P->fasts && !P->fast4096 && !P->fastm && (d.out_mode == XRFTHIP_OUT_POWER || d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_CROSS) && !(d.flags & ~okg) &&
                   !((d.flags & XRFTHIP_HALF_X) && (d.flags & (XRFTHIP_ISO | XRFTHIP_SHIFT_X | XRFTHIP_SHIFT_Y))) &&
                   !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTG", 1) != 0 && fastg_try(P);
        if (P->fastg) {
            int rcg = P->dbl ? fastg_setup_t<double>(P) : fastg_setup_t<float>(P);
            std::vector<double> ones((size_t)std::max<long long>(std::max(d.ny, d.nx), P->g_rows), 1.0);
            std::vector<float> onesf(ones.size(), 1.0f);
            if (!rcg) rcg = P->dbl ? P->ones4096.upload(ones.data(), ones.size() * sizeof(double)) : P->ones4096.upload(onesf.data(), onesf.size() * sizeof(float));
            if (rcg) { delete P; return rcg; }
        }
    }
    {   // every other large real slab whose lengths the butterflies factor (the columns: any length, through a chirp convolution): the y-first pipeline with the
        // lengths as data (fastn.h) -- either pass may still be the table kernel of fastm.h when its length is in the table
        const uint32_t shifts = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X, ish = XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X;
        const uint32_t isof = XRFTHIP_ISO | XRFTHIP_NO_SPECTRUM_OUT, halff = XRFTHIP_HALF_X | XRFTHIP_REALDIM_X2;
        const uint32_t allowed = d.out_mode == XRFTHIP_OUT_POWER ? (shifts | isof | halff) : d.out_mode == XRFTHIP_OUT_CROSS ? (shifts | ish | isof | halff)
                                 : (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_PHASE) ? (shifts | ish | XRFTHIP_HALF_X) : 0u;
        const bool half_ok = !((d.flags & halff) && (d.flags & XRFTHIP_ISO)) && !((d.flags & XRFTHIP_HALF_X) && (d.flags & XRFTHIP_SHIFT_X));
        const bool cand = half_ok && d.ndim == 2 && (d.dtype == XRFTHIP_F64 || d.dtype == XRFTHIP_F32) && !P->fast4096 && !P->fastm && !P->fastg && !P->fasts && !(d.flags & ~allowed) &&
                          !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTN", 1) != 0;
        if (cand) {
            P->yny = d.ny; P->ynx = d.nx;
            if (fastn_setup(P)) {
                P->fastm = true; P->yfirst = true;
                const bool two = plan_two(P);
                const int rpu = P->n_r.rt ? P->n_rpu : (two ? fastm_rpu(d.nx, true, P->dbl) : mgeom(d.nx, P->dbl).g);  // (the largest count a row kernel of this plan may use)
                P->y_nrow_pad = (int)((d.ny / 2 + 1 + rpu - 1) / rpu * rpu);
                const long long ylen = P->n_blue_m ? P->n_blue_m : d.ny;
                int rcn = P->dbl ? build_twiddle<double>(P->tw_fx, d.nx, d.nx) : build_twiddle<float>(P->tw_fx, d.nx, d.nx);
                if (!rcn) rcn = P->dbl ? build_twiddle<double>(P->tw_fy, ylen, ylen) : build_twiddle<float>(P->tw_fy, ylen, ylen);
                std::vector<double> ones((size_t)std::max(d.ny, d.nx), 1.0);
                std::vector<float> onesf((size_t)std::max(d.ny, d.nx), 1.0f);
                if (!rcn) rcn = P->dbl ? P->ones4096.upload(ones.data(), ones.size() * sizeof(double)) : P->ones4096.upload(onesf.data(), onesf.size() * sizeof(float));
                if (!rcn && P->n_c.rt && P->n_rad_p) rcn = P->dbl ? fastn_rader_tables<double>(P) : fastn_rader_tables<float>(P);
                else if (!rcn && P->n_c.rt) rcn = P->dbl ? fastn_upload_twm<double>(P->n_c.geo, P->n_c.twm, P->n_blue_m != 0) : fastn_upload_twm<float>(P->n_c.geo, P->n_c.twm, P->n_blue_m != 0);
                if (!rcn && P->n_r.rt) rcn = P->dbl ? fastn_upload_twm<double>(P->n_r.geo, P->n_r.twm) : fastn_upload_twm<float>(P->n_r.geo, P->n_r.twm);
                if (!rcn && P->n_blue_m) rcn = P->dbl ? fastn_blue_tables<double>(P) : fastn_blue_tables<float>(P);
                if (!rcn && P->n_c.rt) rcn = P->n_c.geo_dev.upload(&P->n_c.geo, sizeof(NGeo));
                if (!rcn && P->n_r.rt) rcn = P->n_r.geo_dev.upload(&P->n_r.geo, sizeof(NGeo));
                if (rcn) { delete P; return rcn; }
            }
        }
    }
    if ((d.flags & XRFTHIP_AXIS_Y) && (d.flags & XRFTHIP_PHASE_IN) && !P->fastgy && !P->fastmy) { delete P; return XRFTHIP_BAD_ARG; }  // (the generic column tiles have no input phase)
    if ((d.flags & XRFTHIP_AXIS_Y) && (d.flags & XRFTHIP_HALF_X) && !P->fastgy && !P->fastmy) { delete P; return XRFTHIP_UNSUPPORTED_LENGTH; }  // (... and no half output: the caller transposes)
    set_kernel_attrs_once();
    // nbins must be known before tiles are sized (the LDS histogram shares the tile's allocation): ISO plans are
    // (re)built in xrfthip_plan_set_binmap.  Build now for everything else.
    int rc = XRFTHIP_OK;
    if (!(d.flags & XRFTHIP_ISO)) rc = P->dbl ? build_plan_t<double>(*P) : build_plan_t<float>(*P);
    if (!rc) rc = finalize_plan(P);
    if (rc) { delete P; return rc; }
    *plan = P;
    return XRFTHIP_OK;
}

int xrfthip_plan_destroy(xrfthip_plan* plan) {
    delete plan;
    return XRFTHIP_OK;
}

int xrfthip_plan_set_window(xrfthip_plan* plan, int axis, const double* h_window, int64_t n) {
    if (!plan || axis < 0 || axis > 1) return XRFTHIP_BAD_ARG;
    if (h_window && n != (axis == 0 ? plan->d.ny : plan->d.nx)) return XRFTHIP_BAD_ARG;
    if (plan->sub_x) return xrfthip_plan_set_window(axis == 0 ? plan->sub_y : plan->sub_x, (axis == 1 && plan->sub_x_1d) ? 1 : 0, h_window, n);  // (each one-axis plan transforms its "y"; a 1-D x stage its x)
    if (axis == 0) plan->host_win_y.assign(h_window ? h_window : nullptr, h_window ? h_window + n : nullptr);
    else plan->host_win_x.assign(h_window ? h_window : nullptr, h_window ? h_window + n : nullptr);
    int rc = upload_real_table(plan, plan->win[axis], h_window, n, 0);
    if (!rc) rc = finalize_plan(plan);
    return rc;
}

int xrfthip_plan_set_phase(xrfthip_plan* plan, int axis, const double* h_phase, int64_t n) {
    if (!plan || axis < 0 || axis > 1) return XRFTHIP_BAD_ARG;
    // an input phase of a c2r transform covers the stored half of the x axis only
    const int64_t want = axis == 0 ? plan->d.ny : ((plan->d.flags & XRFTHIP_C2R_X) ? plan->d.nx / 2 + 1 : plan->d.nx);
    if (h_phase && n != want) return XRFTHIP_BAD_ARG;
    if (plan->sub_x) return xrfthip_plan_set_phase(axis == 0 ? plan->sub_y : plan->sub_x, (axis == 1 && plan->sub_x_1d) ? 1 : 0, h_phase, n);
    plan->host_phase[axis].assign(h_phase ? h_phase : nullptr, h_phase ? h_phase + 2 * n : nullptr);
    int rc = upload_real_table(plan, plan->phase[axis], h_phase, n, 1);
    if (!rc) rc = finalize_plan(plan);  // (a non-trivial phase can take an isotropic cross spectrum off the specialised path: new layout)
    return rc;
}

int xrfthip_plan_set_binmap(xrfthip_plan* plan, const int32_t* h_binmap, int64_t ny, int64_t nx_out, int32_t nbins) {
    if (!plan || !h_binmap || !(plan->d.flags & XRFTHIP_ISO)) return XRFTHIP_BAD_ARG;
    if (ny != plan->d.ny || nx_out != plan->nx_out || nbins < 1) return XRFTHIP_BAD_ARG;
    int rc = plan->binmap.upload(h_binmap, (size_t)ny * nx_out * sizeof(int32_t));
    if (rc) return rc;
    plan->nbins = nbins;
    if (plan->fasts) {  // a radial map within the workgroup's reach: the sums are taken from the staged rows (fasts.h); else the other paths
        const int rcs = fasts_build_tfirst(plan, h_binmap);
        if (rcs) return rcs;
    }
    if (plan->fastg) {  // any map: per-bin position lists (fastg.h)
        const int rcs = fastg_build_iso(plan, h_binmap);
        if (rcs) return rcs;
    }
    if (plan->fast4096 && !plan->fasts) {
        int rcf = XRFTHIP_OK;
        if (plan->yfirst) {
            rcf = fasty_build_tcodes(plan, h_binmap);
            if (!rcf && !plan->ytfirst_on && !fasty_iso_tables_fit(plan, nbins)) plan->fast4096 = false;  // (any map: the atomic tables alias half of the transforms' LDS)
        }
        if (rcf) return rcf;
    }
    if (plan->fastm) {  // a radial map: the fused radial sums are gathered per bin (fastm_rows_kernel)
        const int rcf = fastm_build_tfirst(plan, h_binmap);
        if (rcf) return rcf;
    }
    plan->passes.clear();
    plan->passes_f0.clear();
    int rcb = plan->dbl ? build_plan_t<double>(*plan) : build_plan_t<float>(*plan);
    if (!rcb) rcb = finalize_plan(plan);
    return rcb;
}

// The memory floor of the headline path, measured here and now (selftest.h): `reps` rounds of [pass-1 skeleton, pass-2 skeleton] over
// `nslab` 4096 x 4096 float32 slabs and `reps` plain copies of the same input, HIP events on `stream` around every launch.
// Synchronises (it is a measurement, not part of the hot path).  d_in: nslab x 4096 x 4096 float32; d_w2: nslab x 2052 x 4096 complex64
// (scratch); d_out: nslab x 4096 x 4096 float32 (overwritten).  us[0..2] = average microseconds PER SLAB of the copy, the column
// skeleton and the row skeleton.
int xrfthip_selftest_floor(const void* d_in, void* d_w2, void* d_out, int64_t nslab, int32_t reps, double* us, void* stream) {
    if (!d_in || !d_w2 || !d_out || !us || nslab < 1 || nslab > 4096 || reps < 1 || reps > 1000) return XRFTHIP_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    set_kernel_attrs_once();
    const size_t lds_c = ycols_geom(4096).lds, lds_r = yrows_geom(4096).lds;
    const int m = (int)kLdsMax;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&selftest_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&selftest_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    hipEvent_t ev[4];
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    double acc[3] = {0.0, 0.0, 0.0};
    const size_t n16 = (size_t)nslab * 4096 * 4096 / 4;
    int rc = XRFTHIP_OK;
    for (int r = -1; r < reps && rc == XRFTHIP_OK; ++r) {  // (round -1: warm-up, not counted)
        auto kc = &selftest_copy_kernel;
        auto k1 = &selftest_cols_kernel;
        auto k2 = &selftest_rows_kernel;
        (void)hipEventRecord(ev[0], st);
        XRFT_LAUNCH(kc, dim3(2048), dim3(256), 0, st, reinterpret_cast<const F4*>(d_in), reinterpret_cast<F4*>(d_out), n16);
        (void)hipEventRecord(ev[1], st);
        XRFT_LAUNCH(k1, dim3((unsigned)(nslab * (4096 / SelfGeom::CW))), dim3(512), lds_c, st, reinterpret_cast<const float*>(d_in), reinterpret_cast<cf*>(d_w2), (int)nslab);
        (void)hipEventRecord(ev[2], st);
        XRFT_LAUNCH(k2, dim3((unsigned)(nslab * (SelfGeom::NROW_PAD / SelfGeom::RPU))), dim3(512), lds_r, st, reinterpret_cast<const cf*>(d_w2), reinterpret_cast<float*>(d_out), (int)nslab);
        (void)hipEventRecord(ev[3], st);
        if (hipEventSynchronize(ev[3]) != hipSuccess || hipGetLastError() != hipSuccess) { rc = XRFTHIP_HIP_ERROR; break; }
        if (r < 0) continue;
        for (int i = 0; i < 3; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            acc[i] += (double)ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) us[i] = acc[i] * 1e3 / ((double)reps * (double)nslab);
    return rc;
}

int xrfthip_plan_set_profiling(xrfthip_plan* plan, int enable) {
    if (!plan) return XRFTHIP_BAD_ARG;
    if (plan->sub_x) { const int rc = xrfthip_plan_set_profiling(plan->sub_x, enable); return rc ? rc : xrfthip_plan_set_profiling(plan->sub_y, enable); }
    plan->prof_clear();
    plan->prof_recs.reserve(1 << 16);  // prof_begin hands out pointers into this vector
    plan->prof = enable != 0;
    return XRFTHIP_OK;
}

int xrfthip_plan_profile_read(xrfthip_plan* plan, char* buf, size_t buflen) {
    if (!plan || !buf || !buflen) return XRFTHIP_BAD_ARG;
    if (plan->sub_x) {  // the two one-axis plans' records, one after the other
        const int n1 = xrfthip_plan_profile_read(plan->sub_x, buf, buflen);
        if (n1 < 0) return n1;
        const int n2 = xrfthip_plan_profile_read(plan->sub_y, buf + n1, buflen - (size_t)n1);
        return n2 < 0 ? n2 : n1 + n2;
    }
    std::vector<std::string> order;
    std::map<std::string, std::pair<long long, double>> agg;
    for (auto& r : plan->prof_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        if (!agg.count(r.label)) order.push_back(r.label);
        agg[r.label].first += 1;
        agg[r.label].second += ms;
    }
    std::string s;
    for (auto& l : order) appendf(s, "%s %lld %.6f\n", l.c_str(), agg[l].first, agg[l].second);
    const size_t n = std::min(buflen - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return (int)n;
}

int xrfthip_plan_uses_bluestein(const xrfthip_plan* plan) {
    if (!plan) return 0;
    if (plan->sub_x) return xrfthip_plan_uses_bluestein(plan->sub_x) || xrfthip_plan_uses_bluestein(plan->sub_y);
    if (plan->fastgy) return plan->gy_blue_m > 0;
    if (plan->fastn) return plan->n_blue_m > 0;
    if (plan->fusedi) return 0;
    if (plan->fastyc || plan->fastg || plan->fasts || plan->fastr || plan->fastmx || plan->fastmy || plan->fastm || plan->fast1d || plan->fast4096) return 0;  // (the generic passes of such a plan never run)
    for (const Pass& ps : plan->passes) if (ps.g.blue_n > 0) return 1;
    for (const Pass& ps : plan->passes_f0) if (ps.g.blue_n > 0) return 1;
    return 0;
}

int xrfthip_plan_kernel_info(const xrfthip_plan* plan, int32_t* kind, int32_t* per_workgroup) {
    if (!plan || !kind || !per_workgroup) return XRFTHIP_BAD_ARG;
    const xrfthip_plan* P = plan;
    const bool two = P->d.out_mode == XRFTHIP_OUT_CROSS || P->d.out_mode == XRFTHIP_OUT_PHASE;
    int k = XRFTHIP_K_GENERIC, n = 0;
    if (P->sub_x) k = XRFTHIP_K_COMPOSITE;
    else if (P->fusedi) { k = XRFTHIP_K_FASTN; n = P->n_cw; }
    else if (P->fastg) { k = P->g_one_d ? XRFTHIP_K_FASTG_ROWS : XRFTHIP_K_FASTG; n = P->g_one_d ? P->g_rows : 1; }
    else if (P->fasts) { k = XRFTHIP_K_FASTS; n = 1; }
    else if (P->fastyc) { k = XRFTHIP_K_FASTY; n = 0; }
    else if (P->fastr) { k = XRFTHIP_K_FASTR; n = 1; }
    else if (P->fastmx) { k = XRFTHIP_K_FASTM_X; const MGeomRt C = mxgeom(P->d.nx, P->dbl); n = (two || (P->cplx_in && !(P->d.flags & XRFTHIP_C2R_X))) ? C.g : 2 * C.g; }
    else if (P->fastgy) { k = P->gy_rows ? XRFTHIP_K_FASTG_ROWS : XRFTHIP_K_FASTG_Y; n = ((P->cplx_in || two) ? 1 : 2) * P->gy_G; }
    else if (P->fastmy) { k = XRFTHIP_K_FASTM_Y; const MGeomRt C = mygeom(P->d.ny, P->dbl); n = ((P->cplx_in || two) ? 1 : 2) * C.g; }
    else if (P->fastm) { k = P->fastn ? XRFTHIP_K_FASTN : XRFTHIP_K_FASTM; n = plan_cw(P); }
    else if (fasty_on(P)) { k = XRFTHIP_K_FASTY; n = 0; }
    *kind = k; *per_workgroup = n;
    return XRFTHIP_OK;
}

size_t xrfthip_workspace_bytes(const xrfthip_plan* plan) {
    if (!plan) return 0;
    return plan->ws_bytes;
}

int xrfthip_plan_describe(const xrfthip_plan* plan, char* buf, size_t buflen) {
    if (!plan || !buf || !buflen) return XRFTHIP_BAD_ARG;
    std::string s;
    const xrfthip_desc& d = plan->d;
    if (plan->fusedi) {
        auto rads = [](const NGeo& g) { std::string t; for (int i = 0; i < g.np; ++i) t += (i ? "x" : "") + std::to_string(g.r[i]); return t; };
        const NGeo &gc = plan->n_c.geo, &gr = plan->n_r.geo;
        appendf(s, "xrfthip plan: [batch %lld][ny %lld][mid %lld][nx %lld][inner %lld] dtype=%d mode=%d detrend=%d flags=0x%x ws=%zuB\n"
                   "  [inner layout] [fastn fused] two passes where the axes lie, no transposed copy: cols: the [ny][mid nx inner] view, %d thr, %d packed column pairs (FFT%d r%s), lds=%zuB -> "
                   "W2[slab][%d/%d][%d][%d][%d] complex -> fit per (slab, inner element) -> rows: %d thr, %d independent elements of one row ky per workgroup (FFT%d r%s), lds=%zuB, plane added "
                   "back in the spectral domain, (ky, kx, e) and its Hermitian twin stored as runs of %d elements\n",
                (long long)d.batch, (long long)d.ny, (long long)plan->mid, (long long)d.nx, (long long)plan->inner, d.dtype, d.out_mode, d.detrend, d.flags, plan->ws_bytes,
                gc.thr, gc.g, gc.n, plan->n_rad_p ? ("Rader, prime " + std::to_string(plan->n_rad_p)).c_str() : rads(gc).c_str(), plan->n_c.lds, plan->y_nrow_pad, plan->n_rk, plan->n_nxb, plan->n_rk, plan->n_cw, gr.thr, gr.g, gr.n, rads(gr).c_str(), plan->n_r.lds, gr.g);
        const size_t n = std::min(buflen - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
        return (int)n;
    }
    if (plan->sub_x) {
        appendf(s, "xrfthip plan: [batch %lld][ny %lld][mid %lld][nx %lld][inner %lld] dtype=%d mode=%d detrend=%d flags=0x%x ws=%zuB\n  [inner layout] no transposed copy: %sx where it lies, then y\n",
                (long long)d.batch, (long long)d.ny, (long long)plan->mid, (long long)d.nx, (long long)plan->inner, d.dtype, d.out_mode, d.detrend, d.flags, plan->ws_bytes,
                d.detrend ? "detrend pass (plane per (batch, inner) element), " : "");
        for (const xrfthip_plan* sp : {plan->sub_x, plan->sub_y}) {
            std::vector<char> tmp(4096);
            xrfthip_plan_describe(sp, tmp.data(), tmp.size());
            s += "  ";
            for (const char* c = tmp.data(); *c; ++c) { s += *c; if (*c == '\n' && c[1]) s += "  "; }
        }
        const size_t n = std::min(buflen - 1, s.size());
        memcpy(buf, s.data(), n);
        buf[n] = 0;
        return (int)n;
    }
    appendf(s, "xrfthip plan: ndim=%d batch=%lld ny=%lld nx=%lld dtype=%d mode=%d detrend=%d flags=0x%x width=%lld nx_out=%lld mirror=%d group=%d ws=%zuB\n",
            d.ndim, (long long)d.batch, (long long)d.ny, (long long)d.nx, d.dtype, d.out_mode, d.detrend, d.flags,
            plan->width, plan->nx_out, (int)plan->mirror, plan->G, plan->ws_bytes);
    if (plan->fastg) {
        std::string rxs, rys;
        for (int r : plan->g_rx) rxs += (rxs.empty() ? "" : "x") + std::to_string(r);
        for (int r : plan->g_ry) rys += (rys.empty() ? "" : "x") + std::to_string(r);
        if (plan->g_one_d)
            appendf(s, "  [fastg rows] one pass, one %d-thread workgroup per %d rows of %lld samples%s: in LDS, radices from the plan (x: %d = %s), a mean / line per row in the "
                       "workgroup, output gathered in output order through the digit-reversal table, lds=%zuB\n",
                    (int)fastg_threads(plan), plan->g_rows, (long long)plan->d.nx, plan->g_packed ? " packed in pairs" : plan->cplx_in ? " (complex input)" : " (an odd length: complex sequences)", plan->g_n,
                    rxs.empty() ? "1" : rxs.c_str(), plan->g_lds);
        else if (plan->g_packed)
            appendf(s, "  [fastg] one pass, one %d-thread workgroup per %lld x %lld slab: the half spectrum (%lld rows of %lld + 1 complex) in LDS, radices from the plan "
                       "(x: %lld = %s on packed rows, y: %lld = %s), exact plane detrend in the workgroup, output gathered in output order through the digit-reversal "
                       "tables, lds=%zuB\n",
                    (int)fastg_threads(plan), (long long)plan->d.ny, (long long)plan->d.nx, (long long)plan->d.ny, (long long)plan->d.nx / 2, (long long)plan->d.nx / 2,
                    rxs.empty() ? "1" : rxs.c_str(), (long long)plan->d.ny, rys.c_str(), plan->g_lds);
        else
            appendf(s, "  [fastg] one pass, one %d-thread workgroup per %lld x %lld slab (complex input or an odd row length: the rows as complex sequences): the spectrum (%lld rows of %lld complex) "
                       "in LDS, radices from the plan (x: %lld = %s, y: %lld = %s), exact plane detrend in the workgroup, output gathered in output order through the "
                       "digit-reversal tables, lds=%zuB\n",
                    (int)fastg_threads(plan), (long long)plan->d.ny, (long long)plan->d.nx, (long long)plan->d.ny, (long long)plan->d.nx, (long long)plan->d.nx,
                    rxs.c_str(), (long long)plan->d.ny, rys.c_str(), plan->g_lds);
        if (plan->d.out_mode == XRFTHIP_OUT_CROSS)
            appendf(s, "  [fastg cross spectrum] both fields' tiles in the workgroup's LDS, F0 conj(F1) on the way out\n");
        if (plan->d.flags & XRFTHIP_ISO)
            appendf(s, "  [fastg radial sums] in the same pass: per bin the LDS positions of its samples (any bin map), a bin per wave, float64, a fixed shuffle tree -- no atomics%s\n",
                    (plan->d.flags & XRFTHIP_NO_SPECTRUM_OUT) ? "; the spectrum is not stored" : "");
    } else if (plan->fasts) {
        const SGeomRt G = sgeom(plan->d.ny, plan->d.nx);
        appendf(s, "  [fasts] one pass, one %d-thread workgroup per %lld x %lld slab (%d fit a CU): the packed columns' transform, their split and the rows' "
                   "transform in registers (32 complex per thread, r32x%lld / r32x%lld, three LDS exchanges in halves), exact plane detrend in the workgroup, |F|^2 "
                   "rows staged in LDS and written whole with the fftshift and the Hermitian mirror, lds=%zuB; 8 algorithmic bytes per sample through memory\n",
                G.thr, (long long)plan->d.ny, (long long)plan->d.nx, G.per_cu, (long long)plan->d.ny / 32, (long long)plan->d.nx / 32, G.lds);
    } else if (plan->fastyc && plan->fastyc_fs) {
        const YGeomRt C = ycols_geom(plan->d.nx / 256), R = yrows_geom(256);
        appendf(s, "  [fasty complex rows, four-step] two passes over the [%lld][256] view of every %lld-point sequence: cols: %d thr, FFT%lld along the view's rows index (input rotation / lag phase / "
                   "conjugation on load) -> W2 in whole lines -> rows: %d thr, %d rows/unit x W_N^(i2 k1), FFT256, stored transposed (X[k1 + %lld k2]: runs of %d samples)%s; 32 bytes per point through memory\n",
                (long long)plan->d.nx / 256, (long long)plan->d.nx, C.thr, (long long)plan->d.nx / 256, R.thr, R.rk, (long long)plan->d.nx / 256, R.rk / 2,
                (plan->d.flags & XRFTHIP_INVERSE) ? "; inverse: conjugate in / out" : "");
    } else if (plan->fastyc) {
        const YGeomRt C = ycols_geom(plan->d.ny), R = yrows_geom(plan->d.nx);
        if (plan->d.flags & XRFTHIP_C2R_X) {
            const YGeomRt R2 = yrows_geom(plan->d.nx / 2);
            appendf(s, "  [fasty complex] cols: %d thr, %d x 2 adjacent complex columns of the half spectrum (FFT%lld, inverse: conjugate in / out) + one block for the Nyquist column, "
                       "%d columns/unit -> W2 -> rows: %d thr, %d rows/unit: the half spectrum of a row back to %lld real samples (FFT%lld on the packed row), whole rows out; "
                       "16 B per point through memory\n", C.thr, C.gxy, (long long)plan->d.ny, 2 * C.gxy, R2.thr, R2.rk, (long long)plan->d.nx, (long long)plan->d.nx / 2);
        } else
        appendf(s, "  [fasty complex] cols: %d thr, %d x 2 adjacent complex columns (FFT%lld, %s), %d columns/unit -> W2[slab][%lld/%d][nx/%d][%d][%d] -> rows: %d thr, %d rows/unit "
                   "(FFT%lld), whole rows out (scale, %sfftshift); 32 B per point through memory\n",
                C.thr, C.gxy, (long long)plan->d.ny, (plan->d.flags & XRFTHIP_INVERSE) ? "inverse: conjugate in / out" : "forward", 2 * C.gxy, (long long)plan->d.ny,
                std::max(1, 16 / (2 * C.gxy)), 2 * C.gxy, std::max(1, 16 / (2 * C.gxy)), 2 * C.gxy, R.thr, R.rk, (long long)plan->d.nx, plan->fph_on ? "phase, " : "");
    } else if (plan->fastr && plan->fastr_rows) {
        const bool c2r = (plan->d.flags & XRFTHIP_C2R_X) != 0;
        const YGeomRt R = yrows_geom(c2r ? plan->d.nx / 2 : plan->d.nx);
        if (c2r) appendf(s, "  [fasty complex rows] one pass: %d thr, %d rows/unit of the row-major half spectrum back to %lld real samples each (FFT%lld on the packed row; two rows per "
                            "thread through one LDS buffer), whole rows out; 8 algorithmic bytes per sample through memory\n", R.thr, R.rk, (long long)plan->d.nx, (long long)plan->d.nx / 2);
        else
        appendf(s, "  [fasty complex rows] one pass: %d thr, %d rows/unit of the row-major input (FFT%lld, %s; two rows per thread through one LDS buffer), whole rows out; "
                   "16 algorithmic bytes per point through memory\n", R.thr, R.rk, (long long)plan->d.nx, (plan->d.flags & XRFTHIP_INVERSE) ? "inverse" : "forward");
    } else if (plan->fastr && plan->fastr_cin) {
        appendf(s, "  [fastr complex rows] one pass, one %lld-thread workgroup per %lld-point complex row: the %s transform in registers (32 per thread, two LDS "
                   "exchanges), natural order through the LDS, lds=%zuB; 16 algorithmic bytes per point through memory\n",
                (long long)plan->d.nx / 32, (long long)plan->d.nx, (plan->d.flags & XRFTHIP_INVERSE) ? "inverse" : "forward",
                plan->d.nx == 16384 ? R2Geom<32, 16>::LDS : plan->d.nx == 8192 ? R2Geom<16, 16>::LDS : plan->d.nx == 4096 ? R2Geom<16, 8>::LDS : R2Geom<8, 8>::LDS);
    } else if (plan->fastr) {
        const long long nxr = plan->d.nx;
        appendf(s, "  [fastr] one pass, one %lld-thread workgroup per %lld-sample row (grid %lld): the packed %lld-point complex transform in registers (32 per thread, "
                   "r32x%dx%d, LDS exchanges%s), real split through the LDS, lds=%zuB; per-row detrend + window + full (or half) spectrum; "
                   "12 algorithmic bytes per sample through memory\n",
                nxr / 64, nxr, plan->tune_rgrid > 0 ? std::min<long long>(plan->tune_rgrid, plan->d.batch) : (long long)plan->d.batch, nxr / 2,
                nxr >= 32768 ? 32 : nxr == 4096 ? 8 : 16, nxr == 65536 ? 32 : nxr <= 8192 ? 8 : 16, nxr == 65536 ? " in halves" : "",
                nxr == 65536 ? kFastRLds : nxr == 32768 ? R2Geom<32, 16>::LDS : nxr == 16384 ? R2Geom<16, 16>::LDS : nxr == 8192 ? R2Geom<16, 8>::LDS : R2Geom<8, 8>::LDS);
    } else if (plan->fastmx) {
        const MGeomRt C = mxgeom(plan->d.nx, plan->dbl);
        appendf(s, "  [fastm x-only] %d thr, %d row pairs per workgroup (FFT%lld r%dx%dx%d in LDS), lds=%zuB: per-row detrend + window + transform + full (or half) spectrum in one pass\n",
                C.thr, C.g, (long long)plan->d.nx, C.r0, C.r1, C.r2, C.lds_cols);
    } else if (plan->fastgy) {
        std::string rys;
        for (int r : plan->g_ry) rys += (rys.empty() ? "" : "x") + std::to_string(r);
        const bool onecol = plan->cplx_in || plan->d.out_mode == XRFTHIP_OUT_CROSS || plan->d.out_mode == XRFTHIP_OUT_PHASE;
        if (plan->gy_rows)
            appendf(s, "  [fastg rows Rader] one pass along the contiguous axis, %d thr, %d sequences (%s) per workgroup, lanes along the samples, lds=%zuB: per-row detrend + window + transform%s\n",
                    plan->gy_thr, plan->gy_G, plan->cplx_in ? "complex rows" : onecol ? "a row of each of the two fields" : "pairs of rows", plan->gy_lds, (plan->d.flags & XRFTHIP_INVERSE) ? "; inverse (conj in, conj out)" : "");
        else
        appendf(s, "  [fastg y-only] one pass, %d thr, %d %s per workgroup (%d bytes of a row), the radices from the plan (y: %lld = %s in LDS), lds=%zuB: "
                   "per-column detrend + window + transform%s, in place in memory order%s\n",
                plan->gy_thr, plan->gy_G, plan->cplx_in ? "complex columns" : onecol ? "columns of each of the two fields" : "packed column pairs",
                (int)((plan->cplx_in ? plan->csize : onecol ? plan->rsize : 2 * plan->rsize) * (size_t)plan->gy_G), (long long)(plan->gy_blue_m ? plan->gy_blue_m : plan->d.ny), rys.c_str(), plan->gy_lds,
                onecol ? "" : " + both columns' spectra", (plan->d.flags & XRFTHIP_INVERSE) ? "; inverse (conj in, conj out)" : "");
        if (plan->gy_rad_p) {
            std::string rps;
            for (int r : plan->gy_rp) rps += (rps.empty() ? "" : "x") + std::to_string(r);
            appendf(s, "  [fastg %s Rader] %lld = %lld x %d: the prime-factor form, no twiddles between the two dimensions; along the prime %d a cyclic convolution of %d = %s points "
                       "(forward passes, * the transformed kernel, inverse passes) inside the tile\n", plan->gy_rows ? "rows:" : "y-only",
                    (long long)plan->gy_n, (long long)(plan->gy_n / plan->gy_rad_p), plan->gy_rad_p, plan->gy_rad_p, plan->gy_rad_p - 1, rps.c_str());
        }
        if (plan->gy_blue_m)
            appendf(s, "  [fastg y-only Bluestein] %lld points as a circular convolution of %d inside the tile (chirp products, forward and inverse passes)%s\n",
                    (long long)plan->d.ny, plan->gy_blue_m, plan->gy_tw_lds ? "" : "; twiddles from memory");
    } else if (plan->fastmy) {
        const MGeomRt C = mygeom(plan->d.ny, plan->dbl);
        appendf(s, "  [fastm y-only] %d thr, %d packed column pairs (FFT%lld r%dx%dx%d in LDS), lds=%zuB: per-column detrend + window + transform + both halves of the spectrum in one pass, in place in memory order\n",
                C.thr, C.g, (long long)plan->d.ny, C.r0, C.r1, C.r2, C.lds_cols);
    } else if (plan->fastm && plan->fastn) {
        auto rads = [](const NGeo& g) { std::string t; for (int i = 0; i < g.np; ++i) t += (i ? "x" : "") + std::to_string(g.r[i]); return t; };
        std::string cs_, rs_;
        if (plan->n_c.rt) {
            const NGeo& g = plan->n_c.geo;
            if (plan->n_rad_p) {
                std::string a, b;
                for (int r : plan->n_rq) a += (a.empty() ? "" : "x") + std::to_string(r);
                for (int r : plan->n_rp) b += (b.empty() ? "" : "x") + std::to_string(r);
                appendf(cs_, "lengths as data, %d thr, %d packed column pairs (FFT%d = %d r%s x prime %d: the prime-factor form, Rader's cyclic convolution of %d = %s points along the prime, in LDS), lds=%zuB",
                        g.thr, g.g, g.n, g.n / plan->n_rad_p, a.c_str(), plan->n_rad_p, plan->n_rad_p - 1, b.c_str(), plan->n_c.lds);
            } else
            appendf(cs_, "lengths as data, %d thr, %d packed column pairs (FFT%d r%s in LDS%s), lds=%zuB", g.thr, g.g, g.n, rads(g).c_str(),
                    plan->n_blue_m ? ": a chirp convolution" : "", plan->n_c.lds);
        } else {
            const MGeomRt C = mgeom_cols(plan->yny, plan->ynx, plan->dbl);
            appendf(cs_, "table kernel, %d thr, %d packed column pairs (FFT%lld r%dx%dx%d)", C.thr, C.g, (long long)plan->yny, C.r0, C.r1, C.r2);
        }
        if (plan->n_r.rt) {
            const NGeo& g = plan->n_r.geo;
            appendf(rs_, "lengths as data, %d thr, %d rows/unit (FFT%d r%s), lds=%zuB", g.thr, plan->n_rpu, g.n, rads(g).c_str(), plan->n_r.lds);
        } else {
            const MGeomRt R = mgeom(plan->ynx, plan->dbl);
            appendf(rs_, "table kernel, %d thr, %d rows/unit (FFT%lld r%dx%dx%d)", R.thr_r1, plan->n_rpu, (long long)plan->ynx, R.r0, R.r1, R.r2);
        }
        appendf(s, "  [fastn] cols: %s -> W2[slab][%d/%d][%d][%d][%d] complex -> fit -> rows: %s, trend added back in the spectral domain, fftshift + mirror rows\n",
                cs_.c_str(), plan->y_nrow_pad, plan->n_rk, plan->n_nxb, plan->n_rk, plan->n_cw, rs_.c_str());
        if (plan->n_blue_m)
            appendf(s, "  [fastn Bluestein] the %lld-point columns as a circular convolution of %d inside the tile (chirp products, two forward transforms)\n", (long long)plan->yny, plan->n_blue_m);
        if ((plan->d.flags & XRFTHIP_ISO) && plan->nbins > 0)
            appendf(s, "  [fastn radial sums] %s\n", fastm_iso_gather(plan) ? "fused into the row pass: radial map, per-bin gather from the spectra in LDS, no atomics"
                                                   : fastm_iso_fused(plan) ? "fused into the row pass: int64 fixed-point tables behind the transforms' LDS"
                                                                           : "a pass over the stored spectrum");
    } else if (plan->fastm) {
        const MGeomRt C = mgeom_cols(plan->yny, plan->ynx, plan->dbl), R = mgeom(plan->ynx, plan->dbl);
        appendf(s, "  [fastm] cols: %d thr, %d packed column pairs (FFT%lld r%dx%dx%d in LDS), lds=%zuB -> W2[slab][%d/%d][nx/%d][%d][%d] complex -> fit -> rows: %d thr, %d rows/unit (FFT%lld r%dx%dx%d), lds=%zuB, trend added back in the spectral domain, fftshift + mirror rows\n",
                C.thr, C.g, (long long)plan->yny, C.r0, C.r1, C.r2, C.lds_cols, plan->y_nrow_pad, fastm_rk2(plan->yny, plan->ynx, plan->d.out_mode >= XRFTHIP_OUT_CROSS, plan->dbl), fastm_cw(plan->yny, plan->ynx, plan->dbl), fastm_rk2(plan->yny, plan->ynx, plan->d.out_mode >= XRFTHIP_OUT_CROSS, plan->dbl), fastm_cw(plan->yny, plan->ynx, plan->dbl),
                R.thr_r1, R.g_r1, (long long)plan->ynx, R.r0, R.r1, R.r2, R.lds_r1);
        if ((plan->d.flags & XRFTHIP_ISO) && plan->nbins > 0)
            appendf(s, "  [fastm radial sums] %s\n", fastm_iso_gather(plan) ? "fused into the row pass: radial map, per-bin gather from the spectra in LDS, no atomics"
                                                   : fastm_iso_fused(plan) ? "fused into the row pass: int64 fixed-point tables behind the transforms' LDS"
                                                                           : "a pass over the stored spectrum (the tables do not fit beside the transforms)");
    } else if (fasty_on(plan)) {
        const YGeomRt C = ycols_geom(plan->yny), R = yrows_geom(plan->ynx, plan->fast1d);
        if (plan->fast1d) appendf(s, "  [fasty four-step] %lld samples = [%lld][%lld]: columns = step 1 (half spectrum k1 <= %lld), rows x W_N^(i2 k1) = step 2, transposed stores + Hermitian mirror\n",
                                  (long long)plan->d.nx, (long long)plan->yny, (long long)plan->ynx, (long long)plan->yny / 2);
        appendf(s, "  [fasty] cols: %d thr, %d x 2 packed column pairs (FFT%lld r16x16x%lld, column-local detrend fused), %d columns/unit, lds=%zuB -> W2[slab][%d/%d][nx/%d][2][%d][%d] -> rows: %d thr, %d rows/unit (FFT%lld r16x16x%lld), lds=%zuB, |F|^2 + fftshift + mirror rows\n",
                C.thr, C.gxy, (long long)plan->d.ny, (long long)plan->d.ny / 256, C.cw, C.lds, plan->y_nrow_pad, C.rk, C.cw, C.rk, 2 * C.gxy,
                R.thr, R.rk, (long long)plan->d.nx, (long long)plan->d.nx / 256, R.lds);
        if ((plan->d.flags & XRFTHIP_ISO) && plan->ytcodes.p)
            appendf(s, "  [fasty radial sums] fused into the row pass (runs of equal bins from the staged rows, int64 fixed-point tables), bin codes: %s\n",
                    plan->ytfirst_on ? "radial map: per-bin gather, no atomics" : plan->ytcodes_compact ? "compact (radial map: first bin + step mask per 16 samples)" : "full (4 bytes per sample)");
    }
    describe_passes(s, plan->passes_f0, "f0");
    describe_passes(s, plan->passes, "main");
    const size_t n = std::min(buflen - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
    return (int)n;
}

int xrfthip_exec(const xrfthip_plan* plan, const void* d_in0, const void* d_in1, void* d_out, void* d_iso,
                 void* d_workspace, size_t ws_bytes, void* stream) {
    if (!plan || !d_in0) return XRFTHIP_BAD_ARG;
    const xrfthip_plan* P = plan;
    const xrfthip_desc& d = P->d;
    const bool cross = d.out_mode == XRFTHIP_OUT_CROSS || d.out_mode == XRFTHIP_OUT_PHASE;
    const bool iso = (d.flags & XRFTHIP_ISO) != 0;
    if (cross && !d_in1) return XRFTHIP_BAD_ARG;
    if (!d_out && !(d.flags & XRFTHIP_NO_SPECTRUM_OUT)) return XRFTHIP_BAD_ARG;
    if (iso && (!d_iso || !P->binmap.p)) return d_iso ? XRFTHIP_MISSING_TABLE : XRFTHIP_BAD_ARG;
    if (P->fusedi) {
        if (ws_bytes < P->ws_bytes || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
        return d.batch == 0 ? XRFTHIP_OK : run_fused_inner(P, d_in0, d_in1, d_out, (char*)d_workspace, (hipStream_t)stream);
    }
    if (P->sub_x) {
        if (ws_bytes < P->ws_bytes || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
        return d.batch == 0 ? XRFTHIP_OK : run_inner_plan(P, d_in0, d_out, (char*)d_workspace, (hipStream_t)stream);
    }
    if (P->passes.empty()) return XRFTHIP_MISSING_TABLE;
    if (ws_bytes < P->ws_bytes || (!d_workspace && P->ws_bytes)) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (d.batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)d_workspace;
    void* out = (d.flags & XRFTHIP_NO_SPECTRUM_OUT) ? nullptr : d_out;
    const bool det = d.detrend != XRFTHIP_DETREND_NONE;
    double* acc = (double*)(ws + P->off_acc);
    double* coef = (double*)(ws + P->off_coef);
    if (iso) HIP_TRY(hipMemsetAsync(d_iso, 0, (size_t)d.batch * P->nbins * (cross ? 16 : 8), st));
    if (P->fastg) return run_fastg(P, d_in0, d_in1, out, (double*)d_iso, st);
    if (P->fasts) return run_fasts(P, d_in0, out, (double*)d_iso, st);
    if (P->fastyc) return run_fastyc(P, d_in0, out, ws, st);
    if (P->fastr) return run_fastr(P, d_in0, out, st);
    if (P->fastmx) return run_fastmx(P, d_in0, d_in1, out, st);
    if (P->fastgy) return run_fastgy(P, d_in0, d_in1, out, st);
    if (P->fastmy) return run_fastmy(P, d_in0, d_in1, out, st);
    if (P->fastm) return run_fastm(P, d_in0, d_in1, out, (double*)d_iso, ws, st);
    if (fasty_on(P)) {
        return run_fasty(P, (const float*)d_in0, (const float*)d_in1, out, (double*)d_iso, ws, st);
    }
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        int rc;
        if (cross) {
            if (det) {
                rc = P->dbl ? run_moments<double>(P, d_in0, g0, gc, acc, coef, st) : run_moments<float>(P, d_in0, g0, gc, acc, coef, st);
                if (rc) return rc;
            }
            rc = P->dbl ? run_pipeline<double>(P, P->passes_f0, d_in0, nullptr, nullptr, ws, det ? coef : nullptr, g0, gc, st)
                        : run_pipeline<float>(P, P->passes_f0, d_in0, nullptr, nullptr, ws, det ? coef : nullptr, g0, gc, st);
            if (rc) return rc;
        }
        const void* in_main = cross ? d_in1 : d_in0;
        double* acc_m = cross ? acc + (size_t)P->G * P->mom_chunks * 6 : acc;
        double* coef_m = cross ? coef + d.batch * ((d.flags & XRFTHIP_AXIS_Y) ? d.nx : 1) * 6 : coef;
        if (det) {
            rc = P->dbl ? run_moments<double>(P, in_main, g0, gc, acc_m, coef_m, st) : run_moments<float>(P, in_main, g0, gc, acc_m, coef_m, st);
            if (rc) return rc;
        }
        rc = P->dbl ? run_pipeline<double>(P, P->passes, in_main, out, (double*)d_iso, ws, det ? coef_m : nullptr, g0, gc, st)
                    : run_pipeline<float>(P, P->passes, in_main, out, (double*)d_iso, ws, det ? coef_m : nullptr, g0, gc, st);
        if (rc) return rc;
    }
    return XRFTHIP_OK;
}


}  // extern "C"

// xrft_hip.cpp -- plan builder, pass scheduler and the C ABI of libxrft_hip.so (see include/xrft_hip.h).
//
// A plan is a short list of kernel launches per group of slabs, chosen when the plan is created (xrfthip_plan_describe says which):
//   * generic passes (tile_fft.h), any shape:   [slab_moments -> finalize_coef]  ->  x pass(es)  ->  y pass(es)  [-> radial sums]
//     The x pass reads the user's array (detrend / window / flip / ifftshift fused into its loads) and the last pass writes the
//     user's output (fftshift / phase / scaling / |F|^2 / cross / mirror fused into its stores); the only intermediate is the
//     half spectrum of ONE group of slabs, re-used for every group.
//   * the two-pass y-first pipelines: fasty.h (real float32, powers of two 256 .. 4096; the four-step form for long 1-D
//     sequences) and fastm.h (real float64 / float32 on the lat/lon lengths): columns -> plane fit -> rows;
//   * one-pass kernels for ONE transform axis whose length is in the fastm table (fastm_yonly_kernel / fastm_xonly_kernel).
// Nothing here allocates or synchronises in exec.
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/xrft_hip.h"
#ifndef XRFT_EMULATE
#include <hip/hip_ext.h>
#endif
#include "aux_kernels.h"
#include "fastp2.h"
#include "fasty.h"
#include "fasty_iso.h"
#include "fasty_c2c.h"
#include "selftest.h"
#include "fastm.h"
#include "fastn.h"
#include "fastr.h"
#include "fasts.h"
#include "tile_fft.h"
#include "fastg.h"
#ifdef XRFT_SPLIT_TUS  /* the library built from several translation units: the fasty / fastm kernels are instantiated in inst_g*.cpp */
namespace xrft {
#define XRFT_KW extern template __global__
#include "instances.h"
#undef XRFT_KW
}
#endif

using namespace xrft;

namespace {

thread_local int g_last_hip_error = 0;

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            g_last_hip_error = (int)e_;                 \
            return XRFTHIP_HIP_ERROR;                   \
        }                                               \
    } while (0)

constexpr size_t kLdsMax = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may use all of it
constexpr int kCUs = 256;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int upload(const void* host, size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        if (hipMalloc(&p, n ? n : 16) != hipSuccess) { p = nullptr; return XRFTHIP_ALLOC_FAILED; }
        HIP_TRY(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return XRFTHIP_OK;
    }
    void clear() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};

struct FftTables {  // per FFT length, in the plan's precision
    int n = 0;
    std::vector<int> radix;
    bool generic = false;
    DevBuf tw, rev;
    // Bluestein (a prime factor above XRFTHIP_MAX_RADIX): the LDS transform has blue_m = 2^a 3^b 5^c >= 2n-1 points;
    // blue_c[k] = exp(+i pi k^2 / n), blue_b = FFT_m(chirp kernel) / m in the order the DIF passes leave it
    int blue_m = 0;
    DevBuf blue_c, blue_b;
};

enum BufKind { B_NONE = 0, B_IN, B_W, B_W2, B_F0, B_OUT };

struct Pass {
    TileGeom g{};
    Prologue pr{};
    Epilogue ep{};
    bool first = false, final_ = false, generic = false;
    int path = 0;  // tile_fft_kernel PATH: 0 = general instantiation, 1..4 = lean single-purpose ones
    int threads = 256;
    size_t lds = 0;
    int in_kind = B_NONE, out_kind = B_NONE;
    long long outer_per_slab = 1;  // n_outer = outer_per_slab * slabs in the group
    std::string label;
};

long long env_ll(const char* name, long long dflt) {
    const char* e = getenv(name);
    return e && *e ? atoll(e) : dflt;
}

int factorize(long long n, std::vector<int>& out, bool& generic) {
    // prime factors first: anything above XRFTHIP_MAX_RADIX is refused here (Bluestein takes over, see lds_fft_len)
    std::vector<int> big;  // primes other than 2, 3, 5: O(r^2) butterflies of their own
    long long m = n;
    int c2 = 0, c3 = 0, c5 = 0;
    while (m % 2 == 0) { ++c2; m /= 2; }
    while (m % 3 == 0) { ++c3; m /= 3; }
    while (m % 5 == 0) { ++c5; m /= 5; }
    for (long long p = 7; m > 1; p += 2) {
        if (p * p > m) p = m;
        while (m % p == 0) {
            if (p > XRFTHIP_MAX_RADIX) return XRFTHIP_UNSUPPORTED_LENGTH;
            big.push_back((int)p);
            m /= p;
        }
    }
    generic = !big.empty();
    std::sort(big.begin(), big.end(), [](int a, int b) { return a > b; });
    // 2^a 3^b 5^c into as few passes as possible with the in-register butterflies 16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2
    // (every pass is one trip of every point through LDS): exhaustive search, the exponents are small
    static const int R[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
    static const int E2[] = {4, 0, 2, 1, 0, 3, 1, 0, 2, 0, 1}, E3[] = {0, 1, 1, 0, 2, 0, 1, 0, 0, 1, 0}, E5[] = {0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const bool comp = env_ll("XRFTHIP_COMPOSITE", 1) != 0, r16 = env_ll("XRFTHIP_RADIX16", 1) != 0;
    std::vector<int> best, cur;
    std::function<void(int, int, int, int)> dfs = [&](int a2, int a3, int a5, int from) {
        if (a2 == 0 && a3 == 0 && a5 == 0) {
            if (best.empty() || cur.size() < best.size()) best = cur;
            return;
        }
        if (!best.empty() && cur.size() + 1 >= best.size()) return;
        for (int i = from; i < 11; ++i) {  // non-increasing radices: each multiset once, big radices tried first
            if (E2[i] > a2 || E3[i] > a3 || E5[i] > a5) continue;
            if (!comp && (R[i] == 15 || R[i] == 12 || R[i] == 10 || R[i] == 9 || R[i] == 6)) continue;
            if (!r16 && R[i] == 16) continue;
            cur.push_back(R[i]);
            dfs(a2 - E2[i], a3 - E3[i], a5 - E5[i], i);
            cur.pop_back();
        }
    };
    if (c2 || c3 || c5) dfs(c2, c3, c5, 0);
    // DIF order: odd and composite radices first (their passes then run on lane-contiguous LDS), powers of two last
    std::vector<int> odd, two;
    for (int r : best) ((r & (r - 1)) == 0 ? two : odd).push_back(r);
    out = big;
    out.insert(out.end(), odd.begin(), odd.end());
    out.insert(out.end(), two.begin(), two.end());
    if ((int)out.size() > XRFT_MAX_PASSES) return XRFTHIP_UNSUPPORTED_LENGTH;
    return XRFTHIP_OK;
}

// host-side forward FFT (float64) of any length whose prime factors are small: recursive decimation in time over the
// smallest factor (used once per plan for the Bluestein kernel's spectrum)
static void host_fft_rec(const double* xr, const double* xi, size_t n, size_t stride, double* yr, double* yi) {
    if (n == 1) { yr[0] = xr[0]; yi[0] = xi[0]; return; }
    size_t p = 2;
    while (n % p) ++p;
    const size_t m = n / p;
    std::vector<double> tr(n), ti(n);
    for (size_t r = 0; r < p; ++r) host_fft_rec(xr + r * stride, xi + r * stride, m, stride * p, tr.data() + r * m, ti.data() + r * m);
    const long double w0 = -2.0L * 3.14159265358979323846264338327950288L / (long double)n;
    for (size_t k = 0; k < n; ++k) {
        long double ar = 0.0L, ai = 0.0L;
        const size_t km = k % m;
        for (size_t r = 0; r < p; ++r) {
            const long double a = w0 * (long double)((r * k) % n);
            const long double c = cosl(a), s = sinl(a);
            ar += c * tr[r * m + km] - s * ti[r * m + km];
            ai += c * ti[r * m + km] + s * tr[r * m + km];
        }
        yr[k] = (double)ar; yi[k] = (double)ai;
    }
}
static void host_fft_smooth(std::vector<double>& re, std::vector<double>& im) {
    std::vector<double> yr(re.size()), yi(re.size());
    host_fft_rec(re.data(), im.data(), re.size(), 1, yr.data(), yi.data());
    re.swap(yr); im.swap(yi);
}

// length of the transform actually run in LDS for an n-point sequence: n, or the Bluestein length m = 2^a 3^b 5^c >= 2n - 1 (a
// power of two can be almost twice as long and then misses the LDS) when n has a prime factor the radix passes do not take
// -- or take badly: a prime above kGenericMax = 6 runs through the O(r^2) butterfly with its operands in scratch memory
// ((64, 721, 1440) float32, 721 = 7 x 103: 22.3 ms in the column pass, 2.9 GFFT/s -> 54.7 through Bluestein; 1001 = 7 x 11 x 13: 13 ->
// 26; a lone factor 7 breaks even), so it goes to Bluestein as long as four sequences of m points fit the LDS
// (scripts/prof_primes.py, prof_primes2.py).
long long lds_fft_len(long long n, size_t csize) {
    std::vector<int> r;
    bool g;
    if (n < 2) return n;
    const bool ok = factorize(n, r, g) == XRFTHIP_OK;
    static const long long kGenericMax = env_ll("XRFTHIP_GENERIC_MAX", 6);
    if (ok && (!g || r[0] <= kGenericMax)) return n;  // (g: a prime factor other than 2, 3, 5; those come first in r, largest first)
    long long m = 2 * n - 1;
    for (;; ++m) {
        long long q = m;
        while (q % 2 == 0) q /= 2;
        while (q % 3 == 0) q /= 3;
        while (q % 5 == 0) q /= 5;
        if (q == 1) break;
    }
    if (!ok) return m;
    const size_t per = (size_t)(m + m / 16 + 2) * csize;
    // (a factor above 13 through the O(r^2) butterfly is hopeless -- 1460 = 2 x 2 x 5 x 73 daily samples of four years, float64: 1.0 GFFT/s along the time axis --:
    // Bluestein even when only one or two sequences of m points fit the tile)
    const size_t seqs = r[0] > 13 ? 1 : 4;
    return seqs * per + 4096 <= kLdsMax ? m : n;
}

// host-side radix-2 FFT (float64) for the two 4096-point window spectra the fused detrend needs
static void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * 3.14159265358979323846264338327950288 / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const size_t a = i + k, b = i + k + len / 2;
                const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
    }
}

template <typename T>
int build_tables(FftTables& t, int n_logical) {
    t.n = n_logical;
    const int n = (int)lds_fft_len(n_logical, 2 * sizeof(T));
    t.blue_m = n != n_logical ? n : 0;
    int rc = factorize(n, t.radix, t.generic);
    if (rc) return rc;
    std::vector<C2<T>> tw((size_t)std::max(n, 1));
    for (int k = 0; k < n; ++k) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)n;
        tw[k].re = (T)cosl(a);
        tw[k].im = (T)sinl(a);
    }
    std::vector<unsigned> rev((size_t)std::max(n, 1));
    for (int pos = 0; pos < n; ++pos) {  // frequency held at LDS position `pos` after the DIF passes
        long long L = n, rem = pos, k = 0, mult = 1;
        for (int r : t.radix) {
            const long long m = L / r;
            k += (rem / m) * mult;
            rem %= m;
            mult *= r;
            L = m;
        }
        rev[(size_t)k] = (unsigned)pos;
    }
    rc = t.tw.upload(tw.data(), tw.size() * sizeof(C2<T>));
    if (rc) return rc;
    if (t.blue_m) {
        const long long N = n_logical;
        const long double pi = 3.14159265358979323846264338327950288L;
        std::vector<C2<T>> c((size_t)N);
        std::vector<double> br((size_t)n, 0.0), bi((size_t)n, 0.0);
        for (long long k = 0; k < N; ++k) {
            const long double a = pi * (long double)((k * k) % (2 * N)) / (long double)N;  // k^2 mod 2N keeps the angle small
            const long double cr = cosl(a), ci = sinl(a);
            c[(size_t)k].re = (T)cr; c[(size_t)k].im = (T)ci;
            br[(size_t)k] = (double)cr; bi[(size_t)k] = (double)ci;
            if (k) { br[(size_t)(n - k)] = (double)cr; bi[(size_t)(n - k)] = (double)ci; }
        }
        host_fft_smooth(br, bi);
        std::vector<C2<T>> bh((size_t)n);
        for (int k = 0; k < n; ++k) {
            bh[(size_t)rev[(size_t)k]].re = (T)(br[(size_t)k] / n);
            bh[(size_t)rev[(size_t)k]].im = (T)(bi[(size_t)k] / n);
        }
        rc = t.blue_c.upload(c.data(), c.size() * sizeof(C2<T>));
        if (!rc) rc = t.blue_b.upload(bh.data(), bh.size() * sizeof(C2<T>));
        if (rc) return rc;
    }
    return t.rev.upload(rev.data(), rev.size() * sizeof(unsigned));
}

template <typename T>
int build_twiddle(DevBuf& buf, long long N, long long count) {  // W_N^k, k < count
    std::vector<C2<T>> tw((size_t)count);
    for (long long k = 0; k < count; ++k) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)N;
        tw[(size_t)k].re = (T)cosl(a);
        tw[(size_t)k].im = (T)sinl(a);
    }
    return buf.upload(tw.data(), tw.size() * sizeof(C2<T>));
}

}  // namespace

struct xrfthip_plan {
    xrfthip_desc d{};
    bool dbl = false, cplx_in = false;
    size_t rsize = 4, csize = 8;
    long long nxh = 0, width = 0, nx_out = 0, w_cols = 0;  // w_cols: columns of the row->column intermediate incl. tile padding
    bool mirror = false;
    int G = 1;
    int mom_chunks = 1;  // blocks per slab of the moments pass (partial sums added in order: deterministic)
    std::map<int, FftTables> tables;
    std::vector<DevBuf*> extra;  // r2c / four-step twiddles
    DevBuf win[2], phase[2], binmap;
    int nbins = 0;
    std::vector<Pass> passes;     // main pipeline (field 1 for CROSS)
    std::vector<Pass> passes_f0;  // CROSS: field 0 -> raw F0 buffer
    // workspace layout (byte offsets)
    size_t off_acc = 0, off_coef = 0, off_w = 0, off_w2 = 0, off_f0 = 0, off_pt = 0, off_rowfit = 0, off_corr = 0, off_isopart = 0, off_isotmp = 0, off_rdv = 0, ws_bytes = 0;
    int iso_chunks = 1;  // workgroups per slab of the generic radial-sum pass (partial sums added in order)
    std::string desc_text;
    // specialised path for real float32 slabs whose two lengths are 256 .. 4096 powers of two (fasty.h)
    bool fast4096 = false;  // (the flag keeps its first name: the headline shape is where the path started)
    DevBuf tw_fx, tw_fy, ones4096, fph[2];
    std::vector<double> host_phase[2];  // complex, as handed to xrfthip_plan_set_phase (empty = none)
    // two-pass "y first" pipeline for full power spectra (fasty.h): columns -> [fit] -> rows, no untile pass
    bool yfirst = false;
    // ... and, as the two steps of a four-step transform, one long real sequence per slab: N = yny * ynx samples viewed as
    // a [yny][ynx] slab (fasty.h, FS).  yny / ynx are d.ny / d.nx for the 2-D plans.
    bool fast1d = false;
    bool fastyc = false;  // ... the same two passes for COMPLEX float32 slabs (fasty_c2c.h): xrft.ifft over two axes, xrft.fft of complex data
    // ... and its mixed-radix float64 form (fastm.h): lengths 360 / 720 / 1440
    bool fastm = false;
    // ... and the same pipeline with the LENGTHS AS DATA (fastn.h): either pass (or both) of a `fastm` plan may be the run-time-radix kernel -- every
    // length that is a product of the butterflies 2 ... 20 (7, 11, 13 included), and for the columns any other length through a chirp convolution
    bool fastn = false;
    struct NSide { bool rt = false; NGeo geo{}; size_t lds = 0; DevBuf twm, geo_dev; };
    NSide n_c, n_r;                 // pass 1 (columns, length ny) and pass 2 (rows, length nx)
    int n_cw = 0, n_rk = 1, n_rpu = 0, n_nxb = 0;  // the intermediate's layout: columns per block, rows per line; rows per pass-2 workgroup; column blocks per row
    long long y_pitch = 0;          // complex elements per row of the intermediate (ynx, or n_nxb * n_cw when the last column block is ragged)
    int n_blue_m = 0;               // pass 1 through a chirp convolution of this length
    DevBuf n_bluec, n_blueb;
    int n_rad_p = 0;                // ... or, ny = q p with ONE prime 17 ... 127 whose p - 1 the butterflies factor: the prime-factor form with Rader's algorithm along p
    std::vector<int> n_rq, n_rp;    // the radices of q and of p - 1
    DevBuf n_rgeo, n_radpin, n_radpout, n_radb;
    // ... and xrfthip_desc.inner > 1 (two adjacent transform axes, the independent elements innermost) as the same two passes (fastn.h: fastn_cols_kernel on the
    // [ny][nx inner] view, fastn_fit_inner_kernel, fastn_irows_kernel).  n_c: the ny-point columns of the view; n_r: GE sequences of nx points per row workgroup
    bool fusedi = false;
    int n_dbg = 0, fi_dbg = 0, fi_vec = 1;  // the measuring scripts' ablation switches (XRFTHIP_FASTN_DBG / _FI_DBG / _FI_VEC), read when the plan is made: xrfthip_exec reads no environment
    DevBuf winx_exp;                // the window along x expanded to the view's columns (never null: ones)
    // ... and pass 1 alone for ONE transform axis that is not the contiguous one (XRFTHIP_AXIS_Y, fastm_yonly_kernel)
    bool fastmy = false;
    // ... and the same transform over short contiguous rows packed in pairs (ndim = 1, fastm_xonly_kernel)
    bool fastmx = false;
    // ... and ONE pass for a small real slab of any smooth shape, either precision, held in LDS with run-time radices (fastg.h)
    bool fastg = false;
    std::vector<int> g_rx, g_ry;
    DevBuf g_twx, g_twy, g_twr, g_revx, g_revy, g_isopos, g_isostart;
    std::vector<unsigned> g_hrevx, g_hrevy;  // (host copies: the radial-sum lists are built from them when the bin map arrives)
    bool g_one_d = false;  // ... the same kernel on groups of g_rows ROWS of a 1-D transform along x (no y passes, a mean / line per row)
    int g_rows = 0, g_lpr = 1, g_nred = 0;
    int g_rs = 0, g_n = 0;  // LDS row stride; length of the x transforms: nx / 2 (rows packed in pairs of samples) or nx (an odd nx)
    bool g_packed = true;
    // ... and ONE pass for one transform axis that is not the contiguous one (XRFTHIP_AXIS_Y), any smooth length, real input (fastg.h: fastgy_kernel)
    bool fastgy = false;
    int gy_G = 0, gy_thr = 0, gy_blue_m = 0;  // gy_blue_m: Bluestein inside the tile on blue_m rows (a prime factor of ny with no butterfly)
    int gy_rad_p = 0;                         // ... or, ny = q p with ONE such prime p <= 127 and p - 1 smooth: the prime-factor form with Rader's algorithm along p
    bool gy_rows = false;                     // ... the same form along the CONTIGUOUS axis of a 1-D plan ([batch rows][nx samples]; fastgy_kernel FORM 3): gy_n = nx
    long long gy_n = 0;                       // the transform length of a fastgy plan
    std::vector<int> gy_rp;                   // the radices of p - 1
    DevBuf gy_twp, gy_radb, gy_permin;
    bool gy_tw_lds = true;
    size_t gy_lds = 0;
    DevBuf gy_bluec, gy_blueb;
    size_t g_lds = 0;
    // ... and ONE pass for a small real float32 slab that fits the registers of a CU: 256 x 256 power spectra (fasts.h)
    bool fasts = false;
    DevBuf tw_sy, tw_sx, s_tfirst;
    long long tune_sgrid = -1;    // XRFTHIP_FASTS_GRID: workgroups of the launch (0 = one per slab, the default; else a resident set walking the slabs)
    // ... and ONE pass for a long real float32 row that fits the registers of a CU: 65536 samples per workgroup (fastr.h)
    bool fastr = false;
    bool fastr_rows = false;      // ... complex rows of 256 .. 4096 points: pass 2 of the complex two-pass pipeline on the rows of the input itself (fastyc_rows_kernel, nrows > 0)
    bool fastr_cin = false;       // ... its complex-row form: rows of 2048 .. 16384 complex64 points, forward or inverse (fastc_kernel)
    DevBuf tw_rm, tw_rs, tw_rn;   // W_M^p (p < 1024), W_1024^n (n < 32), W_N^p (p < 1024)
    long long tune_rstagger = 0;  // XRFTHIP_FASTR_STAGGER: classes << 8 | units of 3.4 us between the start of consecutive classes of workgroups (FastR::stagger)
    long long tune_rgrid = 0;     // XRFTHIP_FASTR_GRID: workgroups of the launch (0 = one per row; default: a resident set of one per CU walking the rows)
    bool fph_on = false;  // some entry of the combined phase tables (fph) differs from 1
    long long yny = 0, ynx = 0;
    DevBuf tw_big1d;
    int y_nrow_pad = 0;  // rows ky = 0..ny/2 of the intermediate, rounded up to what one row workgroup covers
    DevBuf ywhat0, ywhat1, ytcodes;
    DevBuf ytfirst, ytwin, ytunits;  // (ytwin: the bins each unit of rows reaches; ytunits: the units that reach each bin)
    bool ytfirst_on = false;       // ... and a radial map's: the radial sums are gathered per bin without atomics (fasty_build_tcodes)
    bool ytcodes_compact = false;  // the bin map has a radial map's structure: 4 bytes per 16 samples (fasty_build_tcodes)
    std::vector<double> host_win_y;
    std::vector<double> host_win_x;  // (four-step 1-D: the window of the whole sequence)
    DevBuf win2d;                    // ... as float32, laid out like the slab [yny][ynx]
    bool fast1d_win = false;         // the four-step plan carries a window: slab-shaped window table, per-column window spectra
    // tuning knobs from the environment, read once when the plan is created (never in xrfthip_exec)
    long long tune_group = 0, tune_fast_group = 0, tune_group_bytes = 512LL << 20, tune_cols_grid = 256, tune_max_grid = 8192;
    long long tune_y = 0;  // XRFTHIP_YTUNE: cache policies / start stagger of the y-first float32 kernels (FastY::tune), fixed at plan creation
    long long tune_isorows = 0;  // XRFTHIP_ISOROWS: 1 the persistent radial-sum row kernel (fasty_iso.h) when nothing but the sums leaves pass 2; 0 (default) fasty_rows_kernel<.., ISO>;
                                 // 2 its profiling build (per-phase shader-clock sums printed after every launch: scripts/prof.py iso-phases -- synchronises, never in the product)
    mutable DevBuf iso_tim;      // ... whose counters live here
    // optional per-pass event timing (bench only; a plan with profiling on is not re-entrant)
    bool prof = false;
    struct ProfRec { std::string label; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    void prof_clear() { for (auto& r : prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } prof_recs.clear(); }
    // xrfthip_desc.inner > 1: [batch][ny][nx][inner], two adjacent transform axes with the independent elements innermost.  A
    // composite of two in-place one-axis plans (XRFTHIP_AXIS_Y): sub_x transforms x of [batch ny][nx][inner], sub_y transforms y of
    // [batch][ny][nx inner]; a detrend runs first as a pass of its own (plane_inner_* kernels).  No transposed copy anywhere.
    long long inner = 1, mid = 1;
    bool sub_x_1d = false;  // the x stage is a 1-D plan (nothing behind x: inner = 1), its axis is 1
    xrfthip_plan* sub_x = nullptr;
    xrfthip_plan* sub_y = nullptr;
    size_t off_sub = 0, off_det = 0, off_mid = 0, off_dws = 0;
    ~xrfthip_plan() { for (auto* b : extra) delete b; prof_clear(); delete sub_x; delete sub_y; }
};

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
    // (the one fastm instantiation above 64 KB of dynamic LDS: 4096-point float64 rows, one pair per workgroup)
#define YA_(TT, NN) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, m); \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, m); \
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_yonly_kernel<TT, NN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    YA_(float, 4096); YA_(double, 2048); YA_(double, 4096);
#undef YA_
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fastm_xonly_kernel<double, 4096, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, m);
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
#undef SETA
#define SETF(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, m)
    SETF((fastg_kernel<float, 0, false>)); SETF((fastg_kernel<float, 1, false>)); SETF((fastg_kernel<double, 0, false>)); SETF((fastg_kernel<double, 1, false>));
    SETF((fastg_kernel<float, 2, false>)); SETF((fastg_kernel<double, 2, false>));
    SETF((fastg_kernel<float, 0, true>)); SETF((fastg_kernel<float, 1, true>)); SETF((fastg_kernel<double, 0, true>)); SETF((fastg_kernel<double, 1, true>));
    SETF((fastgy_kernel<float, 0, 0>)); SETF((fastgy_kernel<float, 1, 0>)); SETF((fastgy_kernel<double, 0, 0>)); SETF((fastgy_kernel<double, 1, 0>));
    SETF((fastgy_kernel<float, 0, 1>)); SETF((fastgy_kernel<float, 1, 1>)); SETF((fastgy_kernel<double, 0, 1>)); SETF((fastgy_kernel<double, 1, 1>));
    SETF((fastgy_kernel<float, 0, 2>)); SETF((fastgy_kernel<float, 1, 2>)); SETF((fastgy_kernel<double, 0, 2>)); SETF((fastgy_kernel<double, 1, 2>));
    SETF((fastgy_kernel<float, 0, 3>)); SETF((fastgy_kernel<float, 1, 3>)); SETF((fastgy_kernel<double, 0, 3>)); SETF((fastgy_kernel<double, 1, 3>));
#define SETN(TT, CC) SETF((fastn_cols_kernel<TT, 0, CC>)); SETF((fastn_cols_kernel<TT, 1, 16>)); SETF((fastn_cols_kernel<TT, 2, 16>)); \
                     SETF((fastn_rows_kernel<TT, 0, false, CC>)); SETF((fastn_rows_kernel<TT, 1, false, CC>)); SETF((fastn_rows_kernel<TT, 1, true, CC>)); SETF((fastn_rows_kernel<TT, 2, false, CC>)); \
                     SETF((fastn_rows_kernel<TT, 2, true, CC>)); SETF((fastn_rows_kernel<TT, 3, false, CC>))
    SETN(float, 16); SETN(float, 20); SETN(double, 16);
    SETF((fastn_irows_kernel<float, 0, 16>)); SETF((fastn_irows_kernel<float, 1, 16>)); SETF((fastn_irows_kernel<float, 0, 20>)); SETF((fastn_irows_kernel<float, 1, 20>));
    SETF((fastn_irows_kernel<double, 0, 16>)); SETF((fastn_irows_kernel<double, 1, 16>));
#undef SETN
    SETF((fasts_power_kernel<8, 8, 0, 0>)); SETF((fasts_power_kernel<8, 4, 0, 0>)); SETF((fasts_power_kernel<4, 8, 0, 0>));
    SETF((fasts_power_kernel<8, 8, 0>)); SETF((fasts_power_kernel<8, 8, 1>)); SETF((fasts_power_kernel<8, 8, 2>));  // (above 64 KB of dynamic LDS)
    SETF((fasts_power_kernel<8, 4, 0>)); SETF((fasts_power_kernel<8, 4, 1>)); SETF((fasts_power_kernel<8, 4, 2>));
    SETF((fasts_power_kernel<4, 8, 0>)); SETF((fasts_power_kernel<4, 8, 1>)); SETF((fasts_power_kernel<4, 8, 2>));
    SETF((fastr_kernel<0, false>)); SETF((fastr_kernel<0, true>)); SETF((fastr_kernel<1, false>)); SETF((fastr_kernel<1, true>));
    SETF((fastc_kernel<32, 16, 0>)); SETF((fastc_kernel<32, 16, 1>)); SETF((fastc_kernel<16, 16, 0>)); SETF((fastc_kernel<16, 16, 1>));
    SETF((fastc_kernel<16, 8, 0>)); SETF((fastc_kernel<16, 8, 1>)); SETF((fastc_kernel<8, 8, 0>)); SETF((fastc_kernel<8, 8, 1>));
    SETF((fastr2_kernel<32, 16, 0, false>)); SETF((fastr2_kernel<32, 16, 0, true>)); SETF((fastr2_kernel<32, 16, 1, false>)); SETF((fastr2_kernel<32, 16, 1, true>));
    SETF((fastr2_kernel<16, 16, 0, false>)); SETF((fastr2_kernel<16, 16, 0, true>)); SETF((fastr2_kernel<16, 16, 1, false>)); SETF((fastr2_kernel<16, 16, 1, true>));
#define SETY(NN) SETF((fasty_cols_kernel<NN, false>)); SETF((fasty_cols_kernel<NN, true>)); SETF((fasty_cols_kernel<NN, false, true>)); SETF((fasty_cols_kernel<NN, true, true>)); SETF((fasty_rows_kernel<NN, 1, false>)); SETF((fasty_rows_kernel<NN, 1, true>)); \
                 SETF((fasty_rows_kernel<NN, 0, false>)); SETF((fasty_rows_kernel<NN, 2, false>)); SETF((fasty_rows_kernel<NN, 2, true>)); SETF((fasty_rows_kernel<NN, 3, false>))
    SETY(4096); SETY(2048); SETY(1024); SETY(512); SETY(256);
#undef SETY
#define SETC(NN) SETF((fastyc_cols_kernel<NN>)); SETF((fastyc_rows_kernel<NN>))
    SETC(4096); SETC(2048); SETC(1024); SETC(512); SETC(256);
#undef SETC
#define SETI(NN) SETF((fasty_isorows_kernel<NN, 1, false>)); SETF((fasty_isorows_kernel<NN, 2, false>)); SETF((fasty_isorows_kernel<NN, 1, true>)); SETF((fasty_isorows_kernel<NN, 2, true>))
    SETI(4096); SETI(2048); SETI(1024);
#undef SETI
    SETF((fasty_rows_kernel<256, 0, false, true>)); SETF((fasty_rows_kernel<256, 1, false, true>));
    SETF((fasty_rows_kernel<256, 0, false, true, true>)); SETF((fasty_rows_kernel<256, 1, false, true, true>));
#undef SETF
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

static int upload_real_table(xrfthip_plan* P, DevBuf& buf, const double* h, int64_t n, int cplx) {
    if (!h) { buf.clear(); return XRFTHIP_OK; }
    const size_t cnt = (size_t)n * (cplx ? 2 : 1);
    if (P->dbl) return buf.upload(h, cnt * sizeof(double));
    std::vector<float> f(cnt);
    for (size_t i = 0; i < cnt; ++i) f[i] = (float)h[i];
    return buf.upload(f.data(), cnt * sizeof(float));
}

static bool fast_on(const xrfthip_plan* P);
static bool fastm_iso_fused(const xrfthip_plan* P);
static bool fastm_iso_gather(const xrfthip_plan* P);
static long long fasty_rows_gx(const xrfthip_plan* P);
static bool fasty_iso_tables_fit(const xrfthip_plan* P, int nbins);
static int fastm_rows_rpu(const xrfthip_plan* P);
static int fastm_gather_rpu(const xrfthip_plan* P);

// workgroups per slab of radial_binsum_det_kernel: chunks of <= 2^17 elements (its int64 sums hold 2^17 values), at most 128
static int iso_chunk_count(long long total) {
    long long c = std::max<long long>(1, std::min<long long>(128, total / 16384));
    while ((total + c - 1) / c > (1LL << 17)) ++c;
    return (int)c;
}
// bins per launch: the int64 sums and the exponent table of a window of bins share 64 KB of LDS
static int iso_bin_window(bool cplx) { return (int)((64 * 1024) / (cplx ? 24 : 16)); }  // (+ 4 bytes per bin: the non-finite flags)

// radial sums of `bc` stored spectra [bc][ny][nxo] (rows / columns rotated by sy / sx) -> iso[bc][nbins (x2)], bit-reproducible
static int run_radial_sums(int32_t dtype, const void* spec, const int32_t* d_binmap, long long bc, long long ny, long long nxo, int sy, int sx,
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

static void layout_workspace(xrfthip_plan* P) {
    const xrfthip_desc& d = P->d;
    if (P->fastr || P->fasts || P->fastg || P->fastgy) { P->G = (int)std::max<long long>(1, std::min<long long>(d.batch, 1 << 30)); P->ws_bytes = 0; return; }  // one pass, registers + LDS: no intermediate
    if (P->fastyc) {  // the tiled intermediate of one group of slabs
        long long G = d.slabs_per_group > 0 ? d.slabs_per_group : (P->tune_fast_group > 0 ? P->tune_fast_group : std::max<long long>(1, (32LL * 4096 * 4096) / (d.ny * d.nx)));
        G = std::max<long long>(1, std::min<long long>(G, std::max<long long>(d.batch, 1)));
        P->G = (int)G;
        P->off_w = 0;
        P->ws_bytes = (((size_t)G * (size_t)d.ny * (size_t)d.nx * sizeof(cf)) + 255) & ~(size_t)255;
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

static xrfthip_plan::ProfRec* prof_begin(const xrfthip_plan* P, const std::string& label, hipStream_t st) {
    if (!P->prof || P->prof_recs.size() + 1 >= P->prof_recs.capacity()) return nullptr;
    xrfthip_plan* M = const_cast<xrfthip_plan*>(P);
    xrfthip_plan::ProfRec r;
    r.label = label;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
    (void)hipEventRecord(r.a, st);
    M->prof_recs.push_back(r);
    return &M->prof_recs.back();
}
static void prof_end(xrfthip_plan::ProfRec* r, hipStream_t st) {
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


// combined per-axis factors of the complex modes: the true-phase table (or 1) times (-1)^k for an ifftshifted input
// (rolling the input by n/2 -- xrft.py:436-441 -- is that sign in the spectrum; in a cross spectrum the two signs cancel)
static int fast_phase_tables(xrfthip_plan* P) {
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

static bool phase_nontrivial(const xrfthip_plan* P) {
    for (int ax = 0; ax < 2; ++ax)
        for (size_t k = 0; k + 1 < P->host_phase[ax].size(); k += 2)
            if (std::fabs(P->host_phase[ax][k] - 1.0) > 1e-12 || std::fabs(P->host_phase[ax][k + 1]) > 1e-12) return true;
    return false;
}

// the specialised path is taken unless an isotropic cross spectrum carries a true-phase factor that is not 1 (two
// fields with different lags): its radial sums would need the factor per sample inside the column pass
static bool fast_on(const xrfthip_plan* P) {
    if (P->fastm) return true;
    if (P->fast1d) return true;  // (a window rides on a slab-shaped table: fasty_window_spectra_1d)
    if (!P->fast4096) return false;
    if (P->d.out_mode == XRFTHIP_OUT_CROSS && (P->d.flags & XRFTHIP_ISO) && phase_nontrivial(P)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// two-pass "y first" pipeline (fasty.h): full float32 power spectra of power-of-two slabs
// ---------------------------------------------------------------------------------------------------------------
struct YGeomRt { int thr, gxy, cw, rk, lbs; size_t lds; };
template <int NY> static YGeomRt ycols_geom_t() {
    typedef YCols<NY> Y;
    return {Y::THR, Y::GY, Y::CW, Y::RK, Y::LBS, (size_t)(Y::GY * YLds<NY, Y::GY>::GSTR + 16 * P2<NY>::R3) * sizeof(cf) + (size_t)(Y::THR / 64) * Y::GY * 8 * sizeof(double)};
}
template <int NX, bool FS = false> static YGeomRt yrows_geom_t() {
    typedef YRows<NX, FS> R;
    return {R::THR, R::GX, 0, R::RPU, 0, (size_t)(R::GX * YLds<NX, R::GX>::GSTR + 16 * P2<NX>::R3) * sizeof(cf)};
}
static int ycols_gstr(long long ny) {  // complex elements of LDS per packed column pair of pass 1 (YLds<NY, GY>::GSTR)
    switch (ny) { case 4096: return YLds<4096, YCols<4096>::GY>::GSTR; case 2048: return YLds<2048, YCols<2048>::GY>::GSTR; case 1024: return YLds<1024, YCols<1024>::GY>::GSTR;
                  case 512: return YLds<512, YCols<512>::GY>::GSTR; default: return YLds<256, YCols<256>::GY>::GSTR; }
}
static YGeomRt ycols_geom(long long ny) {
    switch (ny) { case 4096: return ycols_geom_t<4096>(); case 2048: return ycols_geom_t<2048>(); case 1024: return ycols_geom_t<1024>();
                  case 512: return ycols_geom_t<512>(); default: return ycols_geom_t<256>(); }
}
static YGeomRt yrows_geom(long long nx, bool fs = false) {  // .rk = rows per workgroup; fs: the four-step 1-D form (256-point rows)
    if (fs) return yrows_geom_t<256, true>();
    switch (nx) { case 4096: return yrows_geom_t<4096>(); case 2048: return yrows_geom_t<2048>(); case 1024: return yrows_geom_t<1024>();
                  case 512: return yrows_geom_t<512>(); default: return yrows_geom_t<256>(); }
}
static long long fasty_rows_gx(const xrfthip_plan* P) { return yrows_geom(P->ynx, P->fast1d).gxy; }
static int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Four-step 1-D with a window w[n], n = nx i1 + i2: the slab-shaped float32 window table pass 1 reads, and -- column i2 of the view has
// its own window w[nx i1 + i2] -- the per-column transforms FFT_i1(w) and FFT_i1(w (i1 - ibar)) as tables [i2][k1 < nrow_pad] that carry
// the residual line back in pass 2 (fasty_rows_kernel, W2D).  Host, once per plan: nx transforms of ny points each.
static int fasty_window_spectra_1d(xrfthip_plan* P) {
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
static int fasty_window_spectra(xrfthip_plan* P) {
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
static int build_unit_windows(xrfthip_plan* P, const int32_t* bm, int rpu) {
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
static int fastm_build_tfirst(xrfthip_plan* P, const int32_t* bm) {
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
static int fasty_build_tcodes(xrfthip_plan* P, const int32_t* bm) {
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
static bool fasty_iso_tables_fit(const xrfthip_plan* P, int nbins) {
    const YGeomRt R = yrows_geom(P->ynx);
    const size_t half = (size_t)R.gxy * (size_t)(P->ynx + P->ynx / 16) * 4;  // GX rows of floats = GX / 2 rows of complex
    const size_t hw = P->d.out_mode == XRFTHIP_OUT_CROSS ? 2 : 1;
    return nbins <= 65534 && (size_t)nbins * (8 * hw + 4) <= half;
}

static bool fasty_on(const xrfthip_plan* P) { return P->yfirst && fast_on(P); }

// Workgroups of `kernel` the whole device holds at once (a persistent launch's grid): the occupancy calculator's count per CU times the CUs,
// asked once per kernel.
static long long resident_workgroups(const void* kernel, int threads, size_t lds) {
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

static void fasty_launch_cols(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof) {
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

static void fasty_launch_rows(const xrfthip_plan* P, const FastY& p, long long gc, hipStream_t st, bool prof) {
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
static FastY fasty_params(const xrfthip_plan* P, const float* in, void* out, double* iso, char* ws, long long g0, long long gc, int slot, long long slot_slabs) {
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

static int run_fasty(const xrfthip_plan* P, const float* in, const float* in1, void* out, double* iso, char* ws, hipStream_t st) {
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

// ---------------------------------------------------------------------------------------------------------------
// mixed-radix float64 form of the y-first pipeline (fastm.h)
// ---------------------------------------------------------------------------------------------------------------
struct MGeomRt { int thr, g; size_t lds_cols, lds_rows; int r0, r1, r2; int thr_r1, g_r1; size_t lds_r1; };  // *_r1: pass 2 of one field
template <typename T, int N, int GOV = 0> static MGeomRt mgeom_t() {
    typedef MGeom<T, N, GOV> M;
    typedef typename M::template Rows<M::GR1> R1;
    return {M::THR, M::G, M::LDS, M::LDS_ROWS, M::R0, M::R1, M::R2, R1::THR, M::GR1, R1::LDS};
}
static bool fastm_len(long long n, bool dbl) {
#define X_(NN) if (n == NN) return true;
    XRFT_M_LATLON(X_)
    if (dbl) { XRFT_M_POW2(X_) } else { XRFT_M_F32ONLY(X_) }
#undef X_
    return false;
}
static MGeomRt mgeom(long long n, bool dbl) {
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
static bool fastm_wide(long long ny, long long nx, bool dbl) {
    if (dbl || (nx & 7) != 0) return false;
#define X_(NN) if (ny == NN) return true;
    XRFT_M_WIDE32(X_)
#undef X_
    return false;
}
static MGeomRt mgeom_cols(long long ny, long long nx, bool dbl) {  // geometry of pass 1 of an (ny, nx) slab
    if (fastm_wide(ny, nx, dbl)) {
#define X_(NN) if (ny == NN) return mgeom_t<float, NN, 4>();
        XRFT_M_WIDE32(X_)
#undef X_
    }
    return mgeom(ny, dbl);
}
// layout of the intermediate: CW = 2 G columns of a pass-1 workgroup, RK rows per 128-byte line
static int fastm_cw(long long ny, long long nx, bool dbl) { return 2 * mgeom_cols(ny, nx, dbl).g; }
static int fastm_rk(long long ny, long long nx, bool dbl) { const int lb = fastm_cw(ny, nx, dbl) * (dbl ? 16 : 8); return lb >= 128 ? 1 : 128 / lb; }
static int fastm_rpu(long long nx, bool two, bool dbl) { const MGeomRt r = mgeom(nx, dbl); return two ? r.g / 2 : r.g_r1; }  // 
// rows per line of the intermediate for a (ny, nx) plan: a whole 128-byte line of pass 1's CW columns, but never more rows than
// one pass-2 workgroup owns (long float32 sequences: two per workgroup = 4 columns = 32 bytes per row, pass 2 takes 2 rows -> 64-byte pieces)
static int fastm_rk2(long long ny, long long nx, bool two, bool dbl) { return std::max(1, std::min(fastm_rk(ny, nx, dbl), fastm_rpu(nx, two, dbl))); }
// ... of a plan: the table's geometry, or what fastn_setup chose when either pass runs on the run-time-radix kernels (fastn.h)
static bool plan_two(const xrfthip_plan* P) { return P->d.out_mode == XRFTHIP_OUT_CROSS || P->d.out_mode == XRFTHIP_OUT_PHASE; }
static int plan_cw(const xrfthip_plan* P) { return P->fastn ? P->n_cw : fastm_cw(P->yny, P->ynx, P->dbl); }
static int plan_rk2(const xrfthip_plan* P) { return P->fastn ? P->n_rk : fastm_rk2(P->yny, P->ynx, plan_two(P), P->dbl); }
static int plan_nxb(const xrfthip_plan* P) { return P->fastn ? P->n_nxb : (int)(P->ynx / fastm_cw(P->yny, P->ynx, P->dbl)); }

// radial sums inside pass 2 when the per-bin tables fit behind the transforms' LDS (64 KB of dynamic LDS per workgroup); otherwise
// the spectrum is stored and summed by run_radial_sums
// a radial bin map (fastm_build_tfirst) is gathered per bin without atomics or tables; a cross spectrum with a true-phase factor keeps
// the general path (the factor of a sample and of its Hermitian twin differ)
static bool fastm_iso_gather(const xrfthip_plan* P) {
    return P->fastm && (P->d.flags & XRFTHIP_ISO) && P->nbins >= 1 && P->ytfirst_on && !(P->d.out_mode == XRFTHIP_OUT_CROSS && P->fph_on);
}
static bool fastm_iso_fused(const xrfthip_plan* P) {
    if (!P->fastm || !(P->d.flags & XRFTHIP_ISO) || P->nbins < 1) return false;
    if (fastm_iso_gather(P)) return true;
    if (P->fastn && P->n_r.rt) return false;  // (the run-time-radix row kernel fuses the gather of a radial map only: any other map is summed from the stored spectrum)
    const bool cx = P->d.out_mode == XRFTHIP_OUT_CROSS;
    const MGeomRt R = mgeom(P->ynx, P->dbl);
    return (cx ? R.lds_rows : R.lds_r1) + (size_t)P->nbins * (cx ? 20 : 12) <= 64 * 1024;
}

// copies of the per-bin tables in pass 2 (a power of two <= 8, whatever fits the 64 KB)
static int fastm_iso_ncopy(const xrfthip_plan* P) {
    if (fastm_iso_gather(P)) return 1;
    const bool cx = P->d.out_mode == XRFTHIP_OUT_CROSS;
    const MGeomRt R = mgeom(P->ynx, P->dbl);
    const size_t per = (size_t)P->nbins * (cx ? 20 : 12), room = 64 * 1024 - (cx ? R.lds_rows : R.lds_r1);
    int nc = 1;
    while (nc < 8 && per * (size_t)(2 * nc) <= room) nc *= 2;
    return nc;
}

// rows per pass-2 workgroup of this plan: two fields share a workgroup's sequences (MRowsG in fastm.h)
static int fastm_gather_rpu(const xrfthip_plan* P) { return fastm_rows_rpu(P); }
static int fastm_rows_rpu(const xrfthip_plan* P) {
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
static bool fastn_factor(long long n, int maxr, std::vector<int>& out, int need_last = 0) {  // need_last: the largest radix must reach it (the last pass: one butterfly per thread)
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
static void fastn_geom(long long n, const std::vector<int>& rad, int g, int maxthr, bool blue, NGeo& o, int thr_force = 0, int thr_pref = 0) {
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

template <typename T> static int fastn_upload_twm(const NGeo& g, DevBuf& buf, bool blue = false) {  // W_{L_p}^(j k) at [two[p] + j r[p] + k], p = 1 .. np - 2
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
template <typename T> static int fastn_blue_tables(xrfthip_plan* P) {
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

static size_t fastn_lds(const NGeo& g, size_t csize, bool cols) {
    return ((size_t)g.g * g.str + g.twn) * csize + (cols ? (size_t)(g.thr / 64) * g.g * 4 * sizeof(double) : 0);
}

// Radices and thread count of one transform with g sequences per workgroup.  What counts is how many workgroups a CU keeps resident, and that is set by the
// registers (128 per lane in float32 -> 16 waves per CU, 168 in float64 -> 12): the thread count is a divisor of that budget -- 512 (columns) / 256 (rows) in
// float32, 192 / 256 / 384 in float64; 576- or 320-thread workgroups leave a CU half empty (profiles/r05_fastn_threads.txt) -- and the factorisation is the one
// with the fewest passes whose LAST radix is large enough for one last-pass butterfly per thread at that count (a thread loops over the other passes' butterflies).
static size_t fastn_lds(const NGeo& g, size_t csize, bool cols);
static bool fastn_pick(long long n, int g, bool blue, bool dbl, bool cols, int maxr, int thr_force, NGeo& out) {
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

static bool rader_split(long long n, bool allow17, int& p_out, std::vector<int>& rq, std::vector<int>& rp);
// Decide which kernel runs each pass of a y-first plan on (ny, nx) and the layout of the intermediate between them.  Returns false when the plan stays
// with the other paths (a length the butterflies do not factor and the chirp convolution does not fit, sequences that do not fit the LDS).
static bool fastn_setup(xrfthip_plan* P) {
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

static FastN fastn_wrap(const xrfthip_plan* P, const FastM& m, bool cols) {
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

static void fastn_launch_cols(const xrfthip_plan* P, const FastM& m, hipStream_t st) {
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

static void fastn_launch_rows(const xrfthip_plan* P, const FastM& m, long long gc, bool fused, hipStream_t st) {
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

static FastM fastm_params(const xrfthip_plan* P, const void* in, void* out, char* ws, long long g0, long long gc, int slot, long long slot_slabs) {
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

static void fastm_launch_cols(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st) {
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

static void fastm_launch_rows(const xrfthip_plan* P, const FastM& p, long long gc, hipStream_t st) {
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

static int run_fastm(const xrfthip_plan* P, const void* in, const void* in1, void* out, double* iso, char* ws, hipStream_t st) {
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
static bool fastmy_len(long long n, bool dbl) {
#define X_(NN) if (n == NN) return true;
    XRFT_M_LATLON(X_) XRFT_M_POW2(X_) XRFT_M_YONLY(X_)
    if (!dbl) { XRFT_M_F32ONLY(X_) XRFT_M_F32_1AX(X_) }
#undef X_
    return n == 2048 || n == 4096;
}
template <typename T, int N> static MGeomRt mygeom_t() {  // (the y-only kernel's own geometry: at least two sequences per workgroup)
    typedef typename MYGeom<T, N>::type M;
    return {M::THR, M::G, M::LDS, M::LDS_ROWS, M::R0, M::R1, M::R2, 0, 0, 0};
}
static MGeomRt mygeom(long long n, bool dbl) {
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
static int run_fastmy(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st) {
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
    p.nunits = (int)(d.batch * (d.nx / ((two || P->cplx_in) ? C.g : 2 * C.g)));
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
static bool fastmx_len(long long n, bool dbl) { return fastmy_len(n, dbl); }
static MGeomRt mxgeom(long long n, bool dbl) {
    if (n == 4096) return dbl ? mgeom_t<double, 4096>() : mgeom_t<float, 4096>();
    if (dbl && n == 2048) return mgeom_t<double, 2048>();
    return mygeom(n, dbl);
}
static int run_fastmx(const xrfthip_plan* P, const void* in, const void* in1, void* out, hipStream_t st) {
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
    const int rpw = (two || P->cplx_in) ? C.g : 2 * C.g;
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

// ---------------------------------------------------------------------------------------------------------------
// one pass over small slabs of any smooth shape (fastg.h): the lengths as data
// ---------------------------------------------------------------------------------------------------------------
static int fastg_rev(const std::vector<int>& radix, int n, DevBuf& buf, std::vector<unsigned>& host) {  // rev[k] = position of frequency k after the DIF passes (as build_tables)
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
template <typename T> static int fastg_setup_t(xrfthip_plan* P) {
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
static bool fastg_factor(long long n, std::vector<int>& out) {
    bool gen = false;
    if (factorize(n, out, gen) == XRFTHIP_OK && !gen) return true;
    if (n == 7 || n == 11 || n == 13 || n == 14) { out.assign(1, (int)n); return true; }
    return fastn_factor(n, 16, out);
}
static bool fastg_try(xrfthip_plan* P) {  // can the slab's half spectrum live in the LDS of one workgroup, and are both lengths smooth?
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
static int fastg_build_iso(xrfthip_plan* P, const int32_t* bm) {
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
static bool rader_split(long long n, bool allow17, int& p_out, std::vector<int>& rq, std::vector<int>& rp) {
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

static bool fastgy_try(xrfthip_plan* P, bool rows = false) {
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
static int rader_maps(int n, int p, const std::vector<int>& rq_, const std::vector<int>& rp_, std::vector<unsigned>& pin, std::vector<unsigned>& pout, std::vector<double>& bre, std::vector<double>& bim) {
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
template <typename T> static int fastgy_rader_tables(xrfthip_plan* P) {
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
template <typename T> static int fastn_rader_tables(xrfthip_plan* P) {
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
template <typename T> static int fastgy_blue_tables(xrfthip_plan* P) {
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
static int run_fastgy(const xrfthip_plan* P, const void* in, const void* in_b, void* out, hipStream_t st) {
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
static long long fastg_threads(const xrfthip_plan* P) {
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

static int run_fastg(const xrfthip_plan* P, const void* in, const void* in_b, void* out, double* iso, hipStream_t st) {
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

// one pass over small float32 slabs (fasts.h): resident workgroups walk the slabs
struct SGeomRt { int thr; size_t lds; int per_cu; size_t lds_iso; };
template <int RY, int RX> static SGeomRt sgeom_t() {
    typedef SGeom<RY, RX> G;
    const int by_lds = (int)((160 * 1024) / G::LDS);
    return {G::T, G::LDS, std::max(1, std::min(by_lds, (int)G::PER_CU)), G::LDS_ISO};
}
static SGeomRt sgeom(long long ny, long long nx) {
#define SG_(A, B) if (ny == 32 * A && nx == 32 * B) return sgeom_t<A, B>();
    SG_(2, 2) SG_(2, 4) SG_(2, 8) SG_(4, 2) SG_(4, 4) SG_(4, 8) SG_(8, 2) SG_(8, 4) SG_(8, 8)
#undef SG_
    return {0, 0, 0, 0};
}
// fasts: is the bin map a radial one (see fasts_power_kernel, ISO)?  If so: first[ky][b] = the smallest |kx| <= nx/2 of row ky whose bin is
// >= b (nx/2 + 1 if none), ky <= ny/2, b = 0 .. nbins.  Otherwise the plan leaves the one-pass path.
static int fasts_build_tfirst(xrfthip_plan* P, const int32_t* bm) {
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

static int run_fasts(const xrfthip_plan* P, const void* in, void* out, double* iso, hipStream_t st) {
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
    const long long res = P->tune_sgrid < 0 ? 0 : P->tune_sgrid;
    const long long g = res > 0 ? std::min<long long>(res, d.batch) : d.batch;
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

// the two-pass pipeline on complex float32 slabs (fasty_c2c.h): columns -> rows, group by group
static int run_fastyc(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const YGeomRt C = ycols_geom(d.ny), R = yrows_geom(d.nx);
    const int cw = 2 * C.gxy, rk = std::max(1, 16 / cw);
    const size_t lds_c = (size_t)(C.gxy * (ycols_gstr(d.ny)) + 16 * (d.ny / 256)) * sizeof(cf);
    const size_t out_esz = d.out_mode == XRFTHIP_OUT_POWER ? sizeof(float) : sizeof(cf);
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        FastYC p{};
        p.in = reinterpret_cast<const cf*>(in) + (size_t)g0 * d.ny * d.nx;
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
        p.ny = (int)d.ny; p.nx = (int)d.nx; p.nslab = (int)gc;
        p.l_cw = ilog2i(cw); p.l_rk = ilog2i(rk);
        p.power = d.out_mode == XRFTHIP_OUT_POWER ? 1 : 0;
        p.scale = (float)d.scale;
        xrfthip_plan::ProfRec* rec = prof_begin(P, "fastyc_cols", st);
        const dim3 gridc((unsigned)(gc * (d.nx / cw))), blkc((unsigned)C.thr);
#define YCC_(NN) do { auto k = &fastyc_cols_kernel<NN>; XRFT_LAUNCH(k, gridc, blkc, lds_c, st, p); } while (0)
        if (d.ny == 4096) YCC_(4096); else if (d.ny == 2048) YCC_(2048); else if (d.ny == 1024) YCC_(1024); else if (d.ny == 512) YCC_(512); else YCC_(256);
#undef YCC_
        prof_end(rec, st);
        rec = prof_begin(P, "fastyc_rows", st);
        const dim3 gridr((unsigned)(gc * (d.ny / R.rk))), blkr((unsigned)R.thr);
#define YCR_(NN) do { auto k = &fastyc_rows_kernel<NN>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
        if (d.nx == 4096) YCR_(4096); else if (d.nx == 2048) YCR_(2048); else if (d.nx == 1024) YCR_(1024); else if (d.nx == 512) YCR_(512); else YCR_(256);
#undef YCR_
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

// one pass over 65536-sample float32 rows (fastr.h): a 1024-thread workgroup per row, or a resident set walking the rows
static int run_fastr(const xrfthip_plan* P, const void* in, void* out, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    if (P->fastr_rows) {  // complex rows of 256 .. 4096 points: the row pass of the complex two-pass pipeline on the input's own rows
        const YGeomRt R = yrows_geom(d.nx);
        FastYC p{};
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
        p.l_cw = ilog2i((int)d.nx); p.l_rk = 0;
        p.power = d.out_mode == XRFTHIP_OUT_POWER ? 1 : 0;
        p.scale = (float)d.scale;
        p.nrows = d.batch;
        xrfthip_plan::ProfRec* rec = prof_begin(P, "fastyc_rows", st);
        const dim3 gridr((unsigned)((d.batch + R.rk - 1) / R.rk)), blkr((unsigned)R.thr);
#define YCR_(NN) do { auto k = &fastyc_rows_kernel<NN>; XRFT_LAUNCH(k, gridr, blkr, R.lds, st, p); } while (0)
        if (d.nx == 4096) YCR_(4096); else if (d.nx == 2048) YCR_(2048); else if (d.nx == 1024) YCR_(1024); else if (d.nx == 512) YCR_(512); else YCR_(256);
#undef YCR_
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

// Everything xrfthip_exec needs beyond the caller's buffers is built HERE, when the plan is created or one of its tables is
// set: window spectra and phase tables of the specialised paths (device allocations + blocking copies) and the workspace
// layout.  xrfthip_exec itself takes the plan as const: no allocation, no copy, no synchronisation, no getenv.
extern "C" { static int fusedi_tables(xrfthip_plan* P); }
static int finalize_plan(xrfthip_plan* P) {
    if (P->fusedi) return fusedi_tables(P);  // (its workspace layout does not depend on the tables)
    // the radial sums of a cross spectrum with a true-phase factor that is not 1 (two fields with different lags) need the factor per sample: the other paths
    if (P->fastg && P->d.out_mode == XRFTHIP_OUT_CROSS && (P->d.flags & XRFTHIP_ISO) && phase_nontrivial(P)) P->fastg = false;
    if (P->fastg || P->fastgy) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fasts) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
    } else if (P->fastr || P->fastyc) {
        if (P->d.out_mode != XRFTHIP_OUT_POWER) { const int rc = fast_phase_tables(P); if (rc) return rc; }
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

static size_t detrend_inner_ws(bool cplx, long long batch, long long inner);
static int run_detrend_inner(int32_t dtype, int32_t ndim, long long batch, long long ny, long long nx, long long inner, int32_t kind, const void* in, void* out,
                             char* ws, hipStream_t st, long long mid = 1);

// ---------------------------------------------------------------------------------------------------------------
// xrfthip_desc.inner > 1 as two fused passes (fastn.h, round 5): 16 bytes per sample through memory where the composite of two one-axis plans moves 32
// ---------------------------------------------------------------------------------------------------------------
static int fusedi_tables(xrfthip_plan* P) {  // what depends on the windows / phases: called from finalize_plan
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

static xrfthip_plan* create_fused_inner(const xrfthip_desc& d) {
    if (env_ll("XRFTHIP_NO_FAST", 0) || !env_ll("XRFTHIP_FASTN", 1) || !env_ll("XRFTHIP_FUSED_INNER", 1)) return nullptr;
    // (the independent elements innermost, or between the two axes -- not both)
    if (d.ndim != 2 || !((d.mid <= 1 && d.inner >= 2) || (d.mid >= 2 && d.inner <= 1)) || (d.dtype != XRFTHIP_F32 && d.dtype != XRFTHIP_F64)) return nullptr;
    const bool midlay = d.mid >= 2;
    const long long ne = midlay ? d.mid : d.inner;
    if (d.out_mode != XRFTHIP_OUT_COMPLEX && d.out_mode != XRFTHIP_OUT_POWER) return nullptr;
    const uint32_t ok = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X) : 0u);
    if (d.flags & ~ok) return nullptr;  // (a flipped axis: the composite of one-axis plans)
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
    xrfthip_plan* P = new (std::nothrow) xrfthip_plan();
    if (!P) return nullptr;
    P->d = d;
    P->fusedi = true;
    P->inner = d.inner > 1 ? d.inner : 1; P->mid = midlay ? d.mid : 1;
    P->dbl = dbl; P->cplx_in = false; P->rsize = rs; P->csize = cs;
    P->nx_out = d.nx;
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
    long long Gs = d.slabs_per_group > 0 ? d.slabs_per_group : (long long)std::max<size_t>(1, ((size_t)512 << 20) / std::max<size_t>(slab_w, 1));
    Gs = std::max<long long>(1, std::min<long long>(Gs, std::max<long long>(d.batch, 1)));
    P->G = (int)Gs;
    size_t off = 0;
    P->off_w = off; off = al(off + (size_t)Gs * slab_w);
    P->off_rowfit = off; off = al(off + (size_t)Gs * ncol * 4 * sizeof(double));
    P->off_corr = off; off = al(off + (size_t)Gs * ncol * cs);
    P->ws_bytes = off;
    return P;
}

static int run_fused_inner(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st) {
    const xrfthip_desc& d = P->d;
    const bool midlay = d.mid >= 2;
    const long long ne = midlay ? d.mid : d.inner, ncol = d.nx * ne;
    const int sx = midlay ? 1 : (int)d.inner, se = midlay ? (int)d.nx : 1;
    const size_t out_esz = d.out_mode == XRFTHIP_OUT_POWER ? P->rsize : P->csize;
    for (long long g0 = 0; g0 < d.batch; g0 += P->G) {
        const long long gc = std::min<long long>(P->G, d.batch - g0);
        FastM m{};
        m.in = (const char*)in + (size_t)g0 * d.ny * ncol * P->rsize;
        m.w2 = ws + P->off_w;
        m.tw_y = P->tw_fy.p;
        m.win_y = P->win[0].p ? P->win[0].p : P->ones4096.p;
        m.win_x = P->winx_exp.p;
        m.colfit = reinterpret_cast<double*>(ws + P->off_rowfit);
        m.ny = (int)d.ny; m.nx = (int)ncol; m.nrow_pad = P->y_nrow_pad;
        m.l_cw = ilog2i(P->n_cw); m.l_rk = ilog2i(P->n_rk);
        m.detrend = d.detrend; m.nslab = (int)gc; m.nunits = (int)(gc * P->n_nxb);
        xrfthip_plan::ProfRec* rec = prof_begin(P, "fastn_cols", st);
        fastn_launch_cols(P, m, st);
        prof_end(rec, st);
        if (d.detrend) {
            rec = prof_begin(P, "fastn_fit_inner", st);
            const dim3 grid((unsigned)(gc * ne)), blk(256);
            if (P->dbl) { auto k = &fastn_fit_inner_kernel<double>; XRFT_LAUNCH(k, grid, blk, 3 * 256 * sizeof(double), st, (const double*)m.colfit, (const double*)m.win_x, reinterpret_cast<C2<double>*>(ws + P->off_corr), (int)d.nx, (int)ne, (int)d.ny, (int)d.detrend, sx, se); }
            else { auto k = &fastn_fit_inner_kernel<float>; XRFT_LAUNCH(k, grid, blk, 3 * 256 * sizeof(double), st, (const double*)m.colfit, (const float*)m.win_x, reinterpret_cast<C2<float>*>(ws + P->off_corr), (int)d.nx, (int)ne, (int)d.ny, (int)d.detrend, sx, se); }
            prof_end(rec, st);
        }
        FastNI r{};
        r.w2 = m.w2; r.corr = ws + P->off_corr; r.what0 = P->ywhat0.p; r.what1 = P->ywhat1.p;
        r.tw_x = P->tw_fx.p; r.twm = P->n_r.twm.p; r.g = (NGeoPtr)P->n_r.geo_dev.p;
        r.ph_y = P->fph[0].p; r.ph_x = P->fph[1].p; r.ph_on = (d.out_mode != XRFTHIP_OUT_POWER && P->fph_on) ? 1 : 0;
        r.out = (char*)out + (size_t)g0 * d.ny * ncol * out_esz;
        r.ny = (int)d.ny; r.nx = (int)d.nx; r.inner = (int)ne; r.sx = sx; r.se = se; r.midlay = midlay ? 1 : 0; r.nrow_pad = P->y_nrow_pad; r.pitch = (int)P->y_pitch;
        r.l_cw = m.l_cw; r.l_rk = m.l_rk; r.detrend = d.detrend;
        r.shift_y = (d.flags & XRFTHIP_SHIFT_Y) ? (int)(d.ny / 2) : 0;
        r.shift_x = (d.flags & XRFTHIP_SHIFT_X) ? (int)(d.nx / 2) : 0;
        const NGeo& hg = P->n_r.geo;
        {
            const int vw = (int)(16 / out_esz);
            r.vec = (!midlay && (vw == 1 || (d.inner % vw == 0 && hg.g % vw == 0))) ? 1 : 0;
            if (!P->fi_vec) r.vec = 0;
            r.dbg = P->fi_dbg;
        }
        r.neb = (int)((ne + hg.g - 1) / hg.g);
        r.nunits = (int)(gc * (d.ny / 2 + 1) * r.neb);
        r.scale = d.scale;
        int maxrad = 0;
        for (int i = 0; i < hg.np; ++i) maxrad = std::max(maxrad, hg.r[i]);
        const dim3 grid((unsigned)(8 * ((r.nunits + 7) / 8))), blk((unsigned)hg.thr);
        rec = prof_begin(P, "fastn_irows", st);
#define NI_(TT, CC) do { if (d.out_mode == XRFTHIP_OUT_POWER) { auto k = &fastn_irows_kernel<TT, 1, CC>; XRFT_LAUNCH(k, grid, blk, P->n_r.lds, st, r); } \
                         else { auto k = &fastn_irows_kernel<TT, 0, CC>; XRFT_LAUNCH(k, grid, blk, P->n_r.lds, st, r); } } while (0)
        if (P->dbl) NI_(double, 16); else if (maxrad > 16) NI_(float, 20); else NI_(float, 16);
#undef NI_
        prof_end(rec, st);
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

// composite plan for xrfthip_desc.inner > 1 (see xrfthip_plan::inner)
static int create_inner_plan(xrfthip_plan** plan, const xrfthip_desc& d) {
    if (xrfthip_plan* F = create_fused_inner(d)) { *plan = F; return XRFTHIP_OK; }
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

static int run_inner_plan(const xrfthip_plan* P, const void* in, void* out, char* ws, hipStream_t st) {
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
    if ((d.flags & XRFTHIP_REALDIM_X2) && (!(d.flags & XRFTHIP_HALF_X) || d.out_mode == XRFTHIP_OUT_COMPLEX)) return XRFTHIP_BAD_ARG;
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
        const uint32_t okc = XRFTHIP_SHIFT_Y | XRFTHIP_SHIFT_X | (d.out_mode == XRFTHIP_OUT_COMPLEX ? (XRFTHIP_ISHIFT_Y | XRFTHIP_ISHIFT_X | XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u);
        P->fastyc = d.ndim == 2 && d.dtype == XRFTHIP_C64 && fast_len(d.ny) && fast_len(d.nx) && !d.detrend && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) &&
                    !(d.flags & ~okc) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_FASTYC", 1) != 0;
        if (P->fastyc) {
            std::vector<float> ones((size_t)4096, 1.0f);
            int rcc = build_twiddle<float>(P->tw_fx, d.nx, d.nx);
            if (!rcc) rcc = build_twiddle<float>(P->tw_fy, d.ny, d.ny);
            if (!rcc) rcc = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rcc) { delete P; return rcc; }
            P->yny = d.ny; P->ynx = d.nx;
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
        P->fastr_rows = !P->fastr && d.ndim == 1 && d.dtype == XRFTHIP_C64 && fast_len(d.nx) && !d.detrend && (d.out_mode == XRFTHIP_OUT_COMPLEX || d.out_mode == XRFTHIP_OUT_POWER) &&
                        !(d.flags & ~okc) && !env_ll("XRFTHIP_NO_FAST", 0) && env_ll("XRFTHIP_CROWS", 1) != 0;
        if (P->fastr_rows) {
            P->fastr = true;
            P->fastr_cin = false;
            std::vector<float> ones((size_t)4096, 1.0f);
            int rcr = build_twiddle<float>(P->tw_fx, d.nx, d.nx);
            if (!rcr) rcr = P->ones4096.upload(ones.data(), ones.size() * sizeof(float));
            if (rcr) { delete P; return rcr; }
        } else
        if (P->fastr_cin) {
            P->fastr = true;
            P->tune_rgrid = env_ll("XRFTHIP_FASTR_GRID", 0);
            P->tune_rstagger = 0;
            const long long thr = d.nx / 32;  // threads per row: 32 complex values each
            int rcr = build_twiddle<float>(P->tw_rm, d.nx, thr);
            if (!rcr) rcr = build_twiddle<float>(P->tw_rs, thr, 32);
            if (rcr) { delete P; return rcr; }
        } else
        if (P->fastr) {
            // 65536 samples: one resident workgroup per CU walks the rows (measured: 359 vs 344 GFFT/s for a workgroup per row, profiles/r04_fastr.txt);
            // the shorter rows (several workgroups per CU): a workgroup per row
            P->tune_rgrid = env_ll("XRFTHIP_FASTR_GRID", d.nx == 65536 ? kCUs : 0);
            // two classes of workgroups 10 us apart: dft (1024, 65536) 388 -> 441 GFFT/s, power_spectrum 517 -> 586 (profiles/r06_c2_stagger.txt)
            P->tune_rstagger = d.nx == 65536 ? env_ll("XRFTHIP_FASTR_STAGGER", (2 << 8) | 3) : 0;
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
        if (P->fastmy && d.nx % (((two || cplx_in) ? 1 : 2) * mygeom(d.ny, P->dbl).g) != 0) P->fastmy = false;
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
                                 ((cplx_in && d.out_mode == XRFTHIP_OUT_COMPLEX) ? (XRFTHIP_INVERSE | XRFTHIP_PHASE_IN) : 0u);  // (xrft.ifft along the contiguous axis: conj in, conj out, the input rotated)
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
    else if (P->fastmx) { k = XRFTHIP_K_FASTM_X; const MGeomRt C = mxgeom(P->d.nx, P->dbl); n = (two || P->cplx_in) ? C.g : 2 * C.g; }
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
    } else if (plan->fastyc) {
        const YGeomRt C = ycols_geom(plan->d.ny), R = yrows_geom(plan->d.nx);
        appendf(s, "  [fasty complex] cols: %d thr, %d x 2 adjacent complex columns (FFT%lld, %s), %d columns/unit -> W2[slab][%lld/%d][nx/%d][%d][%d] -> rows: %d thr, %d rows/unit "
                   "(FFT%lld), whole rows out (scale, %sfftshift); 32 B per point through memory\n",
                C.thr, C.gxy, (long long)plan->d.ny, (plan->d.flags & XRFTHIP_INVERSE) ? "inverse: conjugate in / out" : "forward", 2 * C.gxy, (long long)plan->d.ny,
                std::max(1, 16 / (2 * C.gxy)), 2 * C.gxy, std::max(1, 16 / (2 * C.gxy)), 2 * C.gxy, R.thr, R.rk, (long long)plan->d.nx, plan->fph_on ? "phase, " : "");
    } else if (plan->fastr && plan->fastr_rows) {
        const YGeomRt R = yrows_geom(plan->d.nx);
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
        return d.batch == 0 ? XRFTHIP_OK : run_fused_inner(P, d_in0, d_out, (char*)d_workspace, (hipStream_t)stream);
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

// per-chunk partial sums of at most 32768 slabs at a time (slabs x chunks <= 32768, 8 sums each) + 8 coefficients per slab
static constexpr long long kDetrendPart = 32768;
size_t xrfthip_detrend_workspace_bytes(int64_t batch) { return ((size_t)(kDetrendPart + std::max<int64_t>(batch, 1)) * 8 * sizeof(double) + 255) & ~(size_t)255; }

int xrfthip_detrend(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int32_t detrend_type,
                    const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1) return XRFTHIP_BAD_ARG;
    if ((ndim != 1 && ndim != 2) || (ndim == 1 && ny != 1)) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (ws_bytes < xrfthip_detrend_workspace_bytes(batch) || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    double* acc = (double*)d_workspace;
    double* coef = acc + kDetrendPart * 8;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long total = ny * nx;
    const size_t esz = (dbl ? 8 : 4) * (cplx ? 2 : 1);
    for (long long b0 = 0; b0 < batch; b0 += 32768) {  // grid.y limit
        const long long bc = std::min<long long>(32768, batch - b0);
        const long long chunks = std::max<long long>(1, std::min<long long>(std::min<long long>(ny, 64), kDetrendPart / bc));
        const dim3 grid((unsigned)chunks, (unsigned)bc), block(256);
        const void* src = (const char*)d_in + (size_t)b0 * total * esz;
        void* dst = (char*)d_out + (size_t)b0 * total * esz;
        const size_t lds = 6 * 256 * sizeof(double);
#define MOM(TT, CC) do { auto k = &slab_moments_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)ny, (long long)nx, total, (long long)nx, acc); } while (0)
        if (dbl) { if (cplx) MOM(double, true); else MOM(double, false); } else { if (cplx) MOM(float, true); else MOM(float, false); }
#undef MOM
        auto kf = &finalize_coef_kernel;
        XRFT_LAUNCH(kf, dim3((unsigned)bc), dim3(64), 0, st, (const double*)acc, coef + b0 * 6, bc, (long long)ny, (long long)nx, (int)detrend_type, (int)chunks);
        const long long gx = std::max<long long>(1, std::min<long long>(2048, (total + 255) / 256));
        const dim3 grid2((unsigned)gx, (unsigned)bc);
#define APP(TT, CC) do { auto k = &detrend_apply_kernel<TT, CC>; XRFT_LAUNCH(k, grid2, block, 0, st, src, dst, (long long)ny, (long long)nx, (const double*)(coef + b0 * 6)); } while (0)
        if (dbl) { if (cplx) APP(double, true); else APP(double, false); } else { if (cplx) APP(float, true); else APP(float, false); }
#undef APP
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

int xrfthip_detrend3(int32_t dtype, int64_t batch, int64_t n0, int64_t n1, int64_t n2, int32_t detrend_type,
                     const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || n0 < 1 || n1 < 1 || n2 < 1) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (ws_bytes < xrfthip_detrend_workspace_bytes(batch) || !d_workspace) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    double* acc = (double*)d_workspace;
    double* coef = acc + kDetrendPart * 8;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long rows = n0 * n1, total = rows * n2;
    const size_t esz = (dbl ? 8 : 4) * (cplx ? 2 : 1);
    for (long long b0 = 0; b0 < batch; b0 += 32768) {  // grid.y limit
        const long long bc = std::min<long long>(32768, batch - b0);
        const long long gx = std::max<long long>(1, std::min<long long>(rows, std::min<long long>(4096, kDetrendPart / bc)));
        const dim3 grid((unsigned)gx, (unsigned)bc), block(256);
        const void* src = (const char*)d_in + (size_t)b0 * total * esz;
        void* dst = (char*)d_out + (size_t)b0 * total * esz;
        const size_t lds = 8 * 256 * sizeof(double);
#define MOM(TT, CC) do { auto k = &block3_moments_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, lds, st, src, (long long)n0, (long long)n1, (long long)n2, acc); } while (0)
        if (dbl) { if (cplx) MOM(double, true); else MOM(double, false); } else { if (cplx) MOM(float, true); else MOM(float, false); }
#undef MOM
        auto kf = &finalize_coef3_kernel;
        XRFT_LAUNCH(kf, dim3((unsigned)((bc + 63) / 64)), dim3(64), 0, st, (const double*)acc, coef + b0 * 8, bc, (long long)n0, (long long)n1, (long long)n2, (int)detrend_type, (int)gx);
        const dim3 grid2((unsigned)std::max<long long>(1, std::min<long long>(rows, 4096)), (unsigned)bc);
#define APP(TT, CC) do { auto k = &detrend3_apply_kernel<TT, CC>; XRFT_LAUNCH(k, grid2, block, 0, st, src, dst, (long long)n0, (long long)n1, (long long)n2, (const double*)(coef + b0 * 8)); } while (0)
        if (dbl) { if (cplx) APP(double, true); else APP(double, false); } else { if (cplx) APP(float, true); else APP(float, false); }
#undef APP
        HIP_TRY(hipGetLastError());
    }
    return XRFTHIP_OK;
}

int xrfthip_spectrum_tail(int32_t dtype, int64_t n, const void* d_a, const void* d_b, void* d_out, double scale, void* stream) {
    if (!d_a || !d_out || n < 0 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
#define TAIL(TT, CC) do { auto k = &spectrum_tail_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<TT>*)d_a, (const C2<TT>*)d_b, d_out, (long long)n, scale); } while (0)
    if (dtype == XRFTHIP_C128) { if (d_b) TAIL(double, true); else TAIL(double, false); }
    else { if (d_b) TAIL(float, true); else TAIL(float, false); }
#undef TAIL
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_angle(int32_t dtype, int64_t n, const void* d_a, void* d_out, void* stream) {
    if (!d_a || !d_out || n < 0 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
    if (dtype == XRFTHIP_C128) { auto k = &angle_kernel<double>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<double>*)d_a, (double*)d_out, (long long)n); }
    else { auto k = &angle_kernel<float>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<float>*)d_a, (float*)d_out, (long long)n); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_spectrum_tail_axis(int32_t dtype, int64_t outer, int64_t na, int64_t inner, int32_t last_is_one, const void* d_a, const void* d_b,
                               void* d_out, double scale, void* stream) {
    if (!d_a || !d_out || outer < 0 || na < 1 || inner < 1 || (dtype != XRFTHIP_C64 && dtype != XRFTHIP_C128)) return XRFTHIP_BAD_ARG;
    const long long n = (long long)outer * na * inner;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
#define TAIL(TT, CC) do { auto k = &spectrum_tail_axis_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, (const C2<TT>*)d_a, (const C2<TT>*)d_b, d_out, n, scale, (long long)na, (long long)inner, (int)last_is_one); } while (0)
    if (dtype == XRFTHIP_C128) { if (d_b) TAIL(double, true); else TAIL(double, false); }
    else { if (d_b) TAIL(float, true); else TAIL(float, false); }
#undef TAIL
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_gather_axis(int32_t elem_bytes, int64_t outer, int64_t n_out, int64_t inner, int64_t n_in, const int64_t* d_index, int64_t roll,
                        const void* d_in, void* d_out, void* stream) {
    if (!d_in || !d_out || d_in == d_out || outer < 0 || n_out < 0 || inner < 0 || n_in < 1) return XRFTHIP_BAD_ARG;
    if (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16) return XRFTHIP_BAD_ARG;
    const long long n = (long long)outer * n_out * inner;
    if (n == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(16384, (n + 255) / 256))), block(256);
    struct alignas(16) E16 { double a, b; };
#define GA(EE) do { auto k = &gather_axis_kernel<EE>; XRFT_LAUNCH(k, grid, block, 0, st, (const EE*)d_in, (EE*)d_out, (long long)outer, (long long)n_out, (long long)inner, (long long)n_in, (const long long*)d_index, (long long)roll); } while (0)
    if (elem_bytes == 4) GA(float); else if (elem_bytes == 8) GA(double); else GA(E16);
#undef GA
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_table_mul(int32_t dtype, int64_t batch, int64_t n_in, int64_t n_out, const void* d_in, const void* d_table, void* d_out, void* stream) {
    if (!d_in || !d_table || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || n_in < 1 || n_out < 1) return XRFTHIP_BAD_ARG;
    if (batch == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long total = batch * n_out;
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 8LL * kCUs * 8)), block(256);
#define TM_(TT, CC) do { auto k = &table_mul_kernel<TT, CC>; XRFT_LAUNCH(k, grid, block, 0, st, d_in, (const C2<TT>*)d_table, (C2<TT>*)d_out, (long long)batch, (long long)n_in, (long long)n_out); } while (0)
    if (dtype == XRFTHIP_F32) TM_(float, false); else if (dtype == XRFTHIP_F64) TM_(double, false); else if (dtype == XRFTHIP_C64) TM_(float, true); else TM_(double, true);
#undef TM_
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

static constexpr int kInnerMaxChunks = 256;
// Row chunks per (batch, inner tile): enough workgroups to fill the chip (a (y, x, t) array is ONE slab), never more than the rows.  The cap
// depends only on (batch, inner) -- what the workspace query knows -- and falls to 1 as soon as the tiles alone fill the chip, so the partial
// sums stay a few MB whatever the inner extent (they were 257 chunks' worth always: 6168 bytes per inner element, 103 GB for a 4096^2 grid).
static int inner_chunk_cap(long long batch, long long i2) {
    const long long tiles = std::max<long long>(1, batch * ((i2 + kInnerThreads - 1) / kInnerThreads));
    return (int)std::max<long long>(1, std::min<long long>(kInnerMaxChunks, (2048 + tiles - 1) / tiles));
}
static int inner_chunks(long long ny, long long batch, long long i2) { return (int)std::max<long long>(1, std::min<long long>(inner_chunk_cap(batch, i2), ny)); }
static size_t detrend_inner_ws(bool cplx, long long batch, long long inner) {
    const size_t i2 = (size_t)inner * (cplx ? 2 : 1);
    return (((size_t)std::max<long long>(batch, 1) * i2 * 3 * sizeof(double) * ((size_t)inner_chunk_cap(batch, (long long)i2) + 1)) + 255) & ~(size_t)255;  // partial sums of <= cap chunks + the coefficients
}
static int run_detrend_inner(int32_t dtype, int32_t ndim, long long batch, long long ny, long long nx, long long inner, int32_t kind, const void* in, void* out,
                             char* ws, hipStream_t st, long long mid) {  // (batch counts (outer, mid) pairs: [batch / mid][ny][mid][nx][inner])
    (void)ndim;
    const bool dbl = dtype == XRFTHIP_F64 || dtype == XRFTHIP_C128, cplx = dtype >= XRFTHIP_C64;
    const long long i2 = inner * (cplx ? 2 : 1);
    const int nch = inner_chunks(ny, batch, i2);
    double* part = reinterpret_cast<double*>(ws);
    double* coef = part + (size_t)batch * inner_chunk_cap(batch, i2) * i2 * 3;
    for (long long b0 = 0; b0 < batch; b0 += 65535) {  // grid.z limit
        const long long bc = std::min<long long>(65535, batch - b0);
        const int ib = (int)std::min<long long>(i2, kInnerThreads), xsn = kInnerThreads / ib;  // lanes across the inner index x column slots
        const dim3 grid((unsigned)((i2 + ib - 1) / ib), (unsigned)nch, (unsigned)bc), block(kInnerThreads);  // (tiles of the inner index on grid.x: no 65535 limit)
        // (the kernel takes the first (outer, mid) pair of the launch and addresses from the array's base: with mid > 1 a block of pairs is not a contiguous piece)
        const size_t lds = (size_t)xsn * 3 * ib * sizeof(double);
        if (dbl) { auto k = &plane_inner_moments_kernel<double>; XRFT_LAUNCH(k, grid, block, lds, st, (const double*)in, (long long)ny, (long long)nx, i2, part, ib, xsn, mid, b0); }
        else { auto k = &plane_inner_moments_kernel<float>; XRFT_LAUNCH(k, grid, block, lds, st, (const float*)in, (long long)ny, (long long)nx, i2, part, ib, xsn, mid, b0); }
    }
    {
        auto k = &plane_inner_finalize_kernel;
        XRFT_LAUNCH(k, dim3((unsigned)((batch * i2 + 255) / 256)), dim3(256), 0, st, (const double*)part, coef, (long long)batch, (long long)ny, (long long)nx, i2, nch, (int)kind);
    }
    const dim3 grid((unsigned)std::min<long long>(batch * ny, 8LL * kCUs * 4)), block(256);
    const size_t clds = (size_t)i2 * 3 * sizeof(double);  // the coefficients of one batch element in LDS (four workgroups per CU at 40 KB)
    const int lds_coef = clds <= 40 * 1024 ? 1 : 0;
    if (dbl) { auto k = &plane_inner_apply_kernel<double>; XRFT_LAUNCH(k, grid, block, lds_coef ? clds : 0, st, (const double*)in, (double*)out, (const double*)coef, (long long)batch, (long long)ny, (long long)nx, i2, lds_coef, mid); }
    else { auto k = &plane_inner_apply_kernel<float>; XRFT_LAUNCH(k, grid, block, lds_coef ? clds : 0, st, (const float*)in, (float*)out, (const double*)coef, (long long)batch, (long long)ny, (long long)nx, i2, lds_coef, mid); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

size_t xrfthip_detrend_inner_workspace_bytes(int32_t dtype, int64_t batch, int64_t inner) {
    if (dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || inner < 1) return 0;
    return detrend_inner_ws(dtype >= XRFTHIP_C64, batch, inner);
}

int xrfthip_detrend_inner(int32_t dtype, int32_t ndim, int64_t batch, int64_t ny, int64_t nx, int64_t inner, int32_t detrend_type,
                          const void* d_in, void* d_out, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || inner < 1) return XRFTHIP_BAD_ARG;
    if ((ndim != 1 && ndim != 2) || (ndim == 1 && ny != 1)) return XRFTHIP_BAD_ARG;
    if (detrend_type != XRFTHIP_DETREND_CONSTANT && detrend_type != XRFTHIP_DETREND_LINEAR) return XRFTHIP_BAD_ARG;
    if (inner > (1LL << 30) || nx > (1LL << 31) - 1 || ny > (1LL << 31) - 1) return XRFTHIP_BAD_ARG;  // (a row's length nx * inner is carried in 64 bits, the positions within it in 32)
    if (!d_workspace || ws_bytes < xrfthip_detrend_inner_workspace_bytes(dtype, batch, inner)) return XRFTHIP_WORKSPACE_TOO_SMALL;
    if (batch == 0) return XRFTHIP_OK;
    return run_detrend_inner(dtype, ndim, batch, ny, nx, inner, detrend_type, d_in, d_out, (char*)d_workspace, (hipStream_t)stream);
}

int xrfthip_convert(int32_t dtype_in, int32_t dtype_out, int64_t n, const void* d_in, void* d_out, void* stream) {
    if (!d_in || !d_out || n < 0) return XRFTHIP_BAD_ARG;
    const bool up = (dtype_in == XRFTHIP_F32 && dtype_out == XRFTHIP_F64) || (dtype_in == XRFTHIP_C64 && dtype_out == XRFTHIP_C128);
    const bool down = (dtype_in == XRFTHIP_F64 && dtype_out == XRFTHIP_F32) || (dtype_in == XRFTHIP_C128 && dtype_out == XRFTHIP_C64);
    if (!up && !down) return XRFTHIP_BAD_ARG;
    const long long cnt = n * (dtype_in >= XRFTHIP_C64 ? 2 : 1);
    if (cnt == 0) return XRFTHIP_OK;
    const dim3 grid((unsigned)std::min<long long>((cnt + 255) / 256, 8LL * kCUs * 4)), block(256);
    if (up) { auto k = &convert_kernel<float, double>; XRFT_LAUNCH(k, grid, block, 0, (hipStream_t)stream, (const float*)d_in, (double*)d_out, cnt); }
    else { auto k = &convert_kernel<double, float>; XRFT_LAUNCH(k, grid, block, 0, (hipStream_t)stream, (const double*)d_in, (float*)d_out, cnt); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

int xrfthip_reduce_axis(int32_t dtype, int64_t outer, int64_t n, int64_t inner, const void* d_in, void* d_out, double scale, void* stream) {
    if (!d_in || !d_out || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || outer < 0 || n < 1 || inner < 0) return XRFTHIP_BAD_ARG;
    if (outer == 0 || inner == 0) return XRFTHIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long in2 = inner * (dtype >= XRFTHIP_C64 ? 2 : 1), total = outer * in2;  // complex data: two real components per sample
    const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 8LL * kCUs * 8)), block(256);
    if (dtype == XRFTHIP_F32 || dtype == XRFTHIP_C64) { auto k = &reduce_axis_kernel<float>; XRFT_LAUNCH(k, grid, block, 0, st, (const float*)d_in, (float*)d_out, (long long)outer, (long long)n, in2, scale); }
    else { auto k = &reduce_axis_kernel<double>; XRFT_LAUNCH(k, grid, block, 0, st, (const double*)d_in, (double*)d_out, (long long)outer, (long long)n, in2, scale); }
    HIP_TRY(hipGetLastError());
    return XRFTHIP_OK;
}

size_t xrfthip_isotropize_workspace_bytes(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, int32_t nbins) {
    if (dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || nbins < 1) return 0;
    return (size_t)batch * iso_chunk_count(ny * nx) * nbins * (dtype >= XRFTHIP_C64 ? 2 : 1) * sizeof(double);
}

int xrfthip_isotropize(int32_t dtype, int64_t batch, int64_t ny, int64_t nx, const void* d_in,
                       const int32_t* d_binmap, int32_t nbins, void* d_iso, void* d_workspace, size_t ws_bytes, void* stream) {
    if (!d_in || !d_binmap || !d_iso || dtype < XRFTHIP_F32 || dtype > XRFTHIP_C128 || batch < 0 || ny < 1 || nx < 1 || nbins < 1) return XRFTHIP_BAD_ARG;
    if (batch == 0) return XRFTHIP_OK;
    if (!d_workspace || ws_bytes < xrfthip_isotropize_workspace_bytes(dtype, batch, ny, nx, nbins)) return XRFTHIP_WORKSPACE_TOO_SMALL;
    return run_radial_sums(dtype, d_in, d_binmap, batch, ny, nx, 0, 0, nbins, iso_chunk_count(ny * nx), (double*)d_workspace, (double*)d_iso, (hipStream_t)stream);
}

}  // extern "C"
